"""Device-side handles: Tanner graph in HBM and decoder workspaces (thin wrappers over the C ABI)."""
import ctypes

import numpy as np

from . import _lib
from .codes import Code


def as_code(obj):
    """Accept a ``codes.Code`` or a dense parity matrix (the reference passes ``code.parity_mtx``, src/biawgn.py:35)."""
    if isinstance(obj, Code):
        return obj
    if hasattr(obj, "edge_chk") and hasattr(obj, "edge_var"):
        return Code.from_edges(obj.m, obj.n, obj.edge_chk, obj.edge_var)
    return Code(None, np.asarray(obj))


def current_device():
    try:
        import torch

        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except ImportError:
        pass
    return 0


def unpack_bits(bits, n, erased=None):
    """Packed decisions (uint32 / int32 words, numpy) -> bytes [B,n] in {0,1} ({0,1,2} with the erased mask), as ldpc_decode returns them."""
    words = np.ascontiguousarray(bits).view(np.uint8)
    x = np.unpackbits(words, axis=1, bitorder="little")[:, :n]
    if erased is not None:
        e = np.unpackbits(np.ascontiguousarray(erased).view(np.uint8), axis=1, bitorder="little")[:, :n]
        x = np.where(e == 1, np.uint8(2), x)
    return x


class CodeHandle:
    def __init__(self, code, device):
        lib = _lib.load()
        self.code, self.device = code, device
        h = ctypes.c_void_p()
        chk = np.ascontiguousarray(code.edge_chk, dtype=np.int32)
        var = np.ascontiguousarray(code.edge_var, dtype=np.int32)
        _lib.check(lib.ldpc_code_create(device, code.m, code.n, code.E, chk.ctypes.data, var.ctypes.data, ctypes.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.load().ldpc_code_destroy(self.h)
                self.h = None
        except Exception:
            pass


def code_handle(code, device=None):
    device = current_device() if device is None else device
    hd = code._handles.get(device)
    if hd is None:
        hd = code._handles[device] = CodeHandle(code, device)
    return hd


class DecoderHandle:
    """One (graph, algorithm, arithmetic, backend) decoder with its HBM workspace."""

    def __init__(self, code, alg, precision="f32", backend="auto", device=None):
        lib = _lib.load()
        self.code_handle = code_handle(code, device)
        self.code, self.alg, self.precision, self.backend = code, alg, precision, backend
        self.np_dtype = np.float64 if precision == "f64" else np.float32
        h = ctypes.c_void_p()
        _lib.check(lib.ldpc_decoder_create(self.code_handle.h, _lib.ALG[alg], _lib.DTYPE[precision], _lib.BACKEND[backend],
                                           ctypes.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.load().ldpc_decoder_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- host (numpy) buffers
    def decode_host(self, priors, y0, max_iter, flags=0):
        lib = _lib.load()
        n = self.code.n
        if self.alg == "BEC":
            y0 = np.ascontiguousarray(np.atleast_2d(y0), dtype=np.uint8)
            B, pri_ptr = y0.shape[0], None
        else:
            priors = np.ascontiguousarray(np.atleast_2d(priors), dtype=self.np_dtype)
            B, pri_ptr = priors.shape[0], priors.ctypes.data
            if y0 is not None:
                y0 = np.ascontiguousarray(np.atleast_2d(y0), dtype=np.uint8)
        if (priors is not None and self.alg != "BEC" and priors.shape[1] != n) or (y0 is not None and y0.shape[1] != n):
            raise ValueError("frames must have n=%d entries" % n)
        xhat = np.empty((B, n), dtype=np.uint8)
        iters = np.empty(B, dtype=np.int32)
        _lib.check(lib.ldpc_decode_host(self.h, pri_ptr, None if y0 is None else y0.ctypes.data, B, int(max_iter), flags,
                                        xhat.ctypes.data, iters.ctypes.data))
        return xhat, iters

    # ---- device (torch) buffers: no copies, current torch stream
    def decode_device(self, priors, y0, max_iter, flags=0, xhat=None, iters=None):
        import torch

        lib = _lib.load()
        n = self.code.n
        ref = y0 if priors is None else priors
        B = ref.shape[0]
        if priors is not None:
            want = torch.float64 if self.precision == "f64" else torch.float32
            if priors.dtype != want or not priors.is_contiguous() or not priors.is_cuda:
                raise ValueError("priors must be a contiguous CUDA tensor of dtype %s" % want)
        if y0 is not None and (y0.dtype != torch.uint8 or not y0.is_contiguous() or not y0.is_cuda):
            raise ValueError("y0 must be a contiguous CUDA uint8 tensor")
        if xhat is None:
            xhat = torch.empty((B, n), dtype=torch.uint8, device=ref.device)
        if iters is None:
            iters = torch.empty((B,), dtype=torch.int32, device=ref.device)
        if B == 0:
            return xhat, iters
        st = torch.cuda.current_stream(ref.device).cuda_stream
        _lib.check(lib.ldpc_decode(self.h, None if priors is None else priors.data_ptr(), None if y0 is None else y0.data_ptr(),
                                   B, int(max_iter), flags, xhat.data_ptr(), iters.data_ptr(), st))
        return xhat, iters

    def decode_device_bits(self, priors, y0, max_iter, flags=0):
        """``ldpc_decode_bits``: packed decisions.  -> (xhat_bits int32 [B, ceil(n/32)] viewed as uint32 words, erased_bits or None, iters);
        bit (v & 31) of word (v >> 5) = decision of variable v.  ``unpack_bits`` expands them."""
        import torch

        ref = y0 if priors is None else priors
        B, W = ref.shape[0], (self.code.n + 31) // 32
        bits = torch.empty((B, W), dtype=torch.int32, device=ref.device)
        era = torch.empty((B, W), dtype=torch.int32, device=ref.device) if self.alg == "BEC" else None
        iters = torch.empty((B,), dtype=torch.int32, device=ref.device)
        if B == 0:
            return bits, era, iters
        st = torch.cuda.current_stream(ref.device).cuda_stream
        _lib.check(_lib.load().ldpc_decode_bits(self.h, None if priors is None else priors.data_ptr(), None if y0 is None else y0.data_ptr(),
                                                B, int(max_iter), flags, bits.data_ptr(), None if era is None else era.data_ptr(),
                                                iters.data_ptr(), st))
        return bits, era, iters

    def decode_host_bits(self, priors, y0, max_iter, flags=0):
        """``ldpc_decode_host_bits``: numpy in, packed decisions out -> (xhat_bits uint32 [B,W], erased_bits uint32 [B,W] or None, iters)."""
        n = self.code.n
        W = (n + 31) // 32
        if self.alg == "BEC":
            y0 = np.ascontiguousarray(np.atleast_2d(y0), dtype=np.uint8)
            B, pri_ptr = y0.shape[0], None
        else:
            priors = np.ascontiguousarray(np.atleast_2d(priors), dtype=self.np_dtype)
            B, pri_ptr = priors.shape[0], priors.ctypes.data
            if y0 is not None:
                y0 = np.ascontiguousarray(np.atleast_2d(y0), dtype=np.uint8)
        bits = np.zeros((B, W), dtype=np.uint32)
        era = np.zeros((B, W), dtype=np.uint32) if self.alg == "BEC" else None
        iters = np.empty(B, dtype=np.int32)
        _lib.check(_lib.load().ldpc_decode_host_bits(self.h, pri_ptr, None if y0 is None else y0.ctypes.data, B, int(max_iter), flags,
                                                     bits.ctypes.data, None if era is None else era.ctypes.data, iters.ctypes.data))
        return bits, era, iters

    def count_errors_bits(self, bits, era, iters, counters, codeword=0, hist_bins=0):
        import torch

        st = torch.cuda.current_stream(counters.device).cuda_stream
        _lib.check(_lib.load().ldpc_count_errors_bits(bits.data_ptr(), None if era is None else era.data_ptr(), None, int(codeword),
                                                      None if iters is None else iters.data_ptr(), bits.shape[0], self.code.n, hist_bins,
                                                      counters.data_ptr(), st))

    def decode_soft_device(self, priors, y0, max_iter, flags=0):
        """Streaming backend with soft output: returns (xhat, iters, marginals) CUDA tensors."""
        import torch

        B, n = priors.shape
        xhat = torch.empty((B, n), dtype=torch.uint8, device=priors.device)
        iters = torch.empty((B,), dtype=torch.int32, device=priors.device)
        marg = torch.zeros_like(priors)
        st = torch.cuda.current_stream(priors.device).cuda_stream
        _lib.check(_lib.load().ldpc_decode_soft(self.h, priors.data_ptr(), None if y0 is None else y0.data_ptr(), B, int(max_iter),
                                                flags, xhat.data_ptr(), iters.data_ptr(), marg.data_ptr(), st))
        return xhat, iters, marg

    def channel_device(self, channel, param, codeword, seed, stream_id, frame0, B, prior_grid=None):
        """Device channel + LLR kernels only: returns (priors or None, y or None) CUDA tensors for frames [frame0, frame0+B).
        ``prior_grid`` = k: the LLRs are rounded to multiples of 2^-k (exact-in-fp32 mode)."""
        import torch

        n = self.code.n
        dt = torch.float64 if self.precision == "f64" else torch.float32
        pri = None if channel == "bec" else torch.empty((B, n), dtype=dt, device="cuda")
        y = None if channel == "biawgn" else torch.empty((B, n), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().ldpc_channel(_lib.CHANNEL[channel] | _lib.ch_prior_grid(prior_grid), _lib.IO_DTYPE[self.precision], float(param), int(codeword), int(seed),
                                            int(stream_id), int(frame0), int(B), n, None if pri is None else pri.data_ptr(),
                                            None if y is None else y.data_ptr(), st))
        return pri, y

    def simulate(self, channel, param, codeword, seed, stream_id, frame0, B, max_iter, counters, flags=0, hist_bins=0):
        """channel -> LLR -> decode -> count for frames [frame0, frame0+B) entirely on the device; ``counters`` is a CUDA
        int64 tensor of 4 + hist_bins entries that is ACCUMULATED into."""
        import torch

        st = torch.cuda.current_stream(counters.device).cuda_stream
        if int(codeword) == -1:
            return self._simulate_random_words(channel, param, seed, stream_id, frame0, B, max_iter, counters, flags, hist_bins, st)
        _lib.check(_lib.load().ldpc_simulate(self.h, _lib.CHANNEL[channel], float(param), int(codeword), int(seed), int(stream_id),
                                             int(frame0), int(B), int(max_iter), flags, hist_bins, counters.data_ptr(), st))

    def simulate_rounds(self, channel, param, codeword, seed, stream_id, frame0, B, rounds, round_stride, max_iter, counters, flags=0, hist_bins=0):
        """``rounds`` passes of ``simulate`` in one call (``ldpc_simulate_rounds``): round r decodes frames frame0 + r * round_stride + [0, B)
        into row r of ``counters`` (CUDA int64 [rounds, 4 + hist_bins], accumulated into).  The LDS-resident erasure decoder runs them as
        ONE launch; every row equals what ``simulate`` gives for that round."""
        import torch

        assert counters.shape == (rounds, 4 + hist_bins) and counters.is_contiguous()
        st = torch.cuda.current_stream(counters.device).cuda_stream
        if int(codeword) == -1:
            for r in range(rounds):
                self._simulate_random_words(channel, param, seed, stream_id, frame0 + r * round_stride, B, max_iter, counters[r], flags, hist_bins, st)
            return
        _lib.check(_lib.load().ldpc_simulate_rounds(self.h, _lib.CHANNEL[channel], float(param), int(codeword), int(seed), int(stream_id),
                                                    int(frame0), int(B), int(rounds), int(round_stride), int(max_iter), flags, hist_bins,
                                                    counters.data_ptr(), st))

    def rounds_per_launch(self):
        """How many Monte-Carlo rounds are worth sending in one ``simulate_rounds`` call: 32 for the LDS-resident erasure decoder (its frame
        positions are refilled across round boundaries; a 65 536-frame round alone is a 0.23 ms launch that mostly ramps up and drains --
        measured at that round size: 286 M frames/s alone, 445 M with 8, 474 M with 16, 490 M with 32 rounds per launch), 1 for everything
        else (their rounds are milliseconds long)."""
        import os

        if self.alg == "BEC" and self.backend != "stream" and self.fused_info()["waves_per_frame"] > 0:
            return int(os.environ.get("LDPC_SIM_ROUNDS_PER_LAUNCH", "32"))
        return 1

    def _simulate_random_words(self, channel, param, seed, stream_id, frame0, B, max_iter, counters, flags, hist_bins, st):
        """``--codeword -1`` (src/main.py:38): every frame sends a random word of the code book (small codes only, as upstream).  A
        composition on the device -- channel kernel (picks the word, adds the noise) -> decode -> count against the sent words."""
        import torch

        lib = _lib.load()
        if getattr(self, "_cb_dev", None) is None:
            cb = getattr(self.code, "cb", None)
            if cb is None:
                raise ValueError("--codeword -1 needs the code book of a small code (Code.cb)")
            self._cb_dev = torch.from_numpy(np.ascontiguousarray(cb, dtype=np.uint8)).cuda()
        n, K = self.code.n, int(self._cb_dev.shape[0])
        dt = torch.float64 if self.precision == "f64" else torch.float32
        step = 1 << 17
        for b0 in range(0, int(B), step):
            nb = min(step, int(B) - b0)
            pri = None if channel == "bec" else torch.empty((nb, n), dtype=dt, device="cuda")
            y = None if channel == "biawgn" else torch.empty((nb, n), dtype=torch.uint8, device="cuda")
            sent = torch.empty((nb, n), dtype=torch.uint8, device="cuda")
            _lib.check(lib.ldpc_channel_words(_lib.CHANNEL[channel], _lib.IO_DTYPE[self.precision], float(param), self._cb_dev.data_ptr(), K,
                                              int(seed), int(stream_id), int(frame0) + b0, nb, n, None if pri is None else pri.data_ptr(),
                                              None if y is None else y.data_ptr(), sent.data_ptr(), st))
            xhat, iters = self.decode_device(pri, y, max_iter, flags)
            _lib.check(lib.ldpc_count_errors_words(xhat.data_ptr(), sent.data_ptr(), iters.data_ptr(), nb, n, hist_bins, counters.data_ptr(), st))

    def fused_info(self):
        out = (ctypes.c_double * 8)()
        _lib.check(_lib.load().ldpc_decoder_fused_info(self.h, out))
        keys = ("waves_per_frame", "lds_gather_cycles_min", "conflict_cycles_identity", "conflict_cycles_planned", "waves_per_cu",
                "lds_bytes_per_frame", "check_rounds", "variable_rounds")
        return dict(zip(keys, list(out)))

    def kernel_name(self, simulate=False):
        """Name of the LDS-resident kernel (as rocprofv3 prints it) this decoder launches; '' on the streaming kernels."""
        buf = ctypes.create_string_buffer(160)
        _lib.check(_lib.load().ldpc_decoder_kernel_name(self.h, 1 if simulate else 0, buf, 160))
        return buf.value.decode()

    def set_profiling(self, on):
        _lib.check(_lib.load().ldpc_decoder_profile(self.h, 1 if on else 0))

    def read_profile(self, reset=True):
        """-> {kernel class: (milliseconds, launches)} accumulated since the last reset (HIP events on the decode stream)."""
        ms = (ctypes.c_double * 4)()
        ln = (ctypes.c_int64 * 4)()
        _lib.check(_lib.load().ldpc_decoder_profile_read(self.h, ms, ln, 1 if reset else 0))
        return {k: (ms[i], ln[i]) for i, k in enumerate(("stream_check_pass", "stream_variable_pass", "fused_decode", "stream_decode_total"))}

    def last_stats(self):
        b, s = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.load().ldpc_decoder_last_stats(self.h, ctypes.byref(b), ctypes.byref(s)))
        return _lib.BACKEND_NAME.get(b.value, "?"), s.value

    def grid_violations(self, reset=True):
        """Exact-in-fp32 mode: (count, global frame indices) of the frames set aside by the exactness guard since the last reset."""
        c = ctypes.c_int64(0)
        frames = np.zeros(4095, dtype=np.int64)
        _lib.check(_lib.load().ldpc_decoder_grid_violations(self.h, ctypes.byref(c), frames.ctypes.data, len(frames), 1 if reset else 0))
        return c.value, frames[:min(c.value, len(frames))].copy()

    def _fp64_sibling(self):
        """The same decoder in the reference's own arithmetic: re-decodes the few frames the exactness guard sets aside."""
        if getattr(self, "_sib64", None) is None:
            self._sib64 = DecoderHandle(self.code, self.alg, "f64", "auto", self.code_handle.device)
        return self._sib64

    def decode_device_exact_fp32(self, priors, max_iter, k):
        """fp32 min-sum on priors that lie on the 2^-k grid (``channel_device(..., prior_grid=k)``), guaranteed to return what the fp64
        reference returns for the same priors: frames whose messages left the range where fp32 sums are exact -- the kernel marks them
        -- are decoded again in fp64.  -> (xhat, iters, frames_redone)"""
        xh, it = self.decode_device(priors, None, max_iter, flags=_lib.flag_prior_grid(k))
        self.grid_violations()
        bad = (it < 0).nonzero().flatten()
        if len(bad):
            x64, i64 = self._fp64_sibling().decode_device(priors[bad].double().contiguous(), None, max_iter)
            xh[bad], it[bad] = x64, i64
        return xh, it, int(len(bad))

    def simulate_exact_fp32(self, param, codeword, seed, stream_id, frame0, B, max_iter, counters, k, hist_bins=0):
        """``simulate`` over BI-AWGN in the exact-in-fp32 mode: the LDS-resident fp32 kernel draws priors on the 2^-k grid and counts every
        frame its guard vouches for; the others (listed by the kernel) are generated again, decoded in fp64 and added to ``counters``
        here.  The counters are then those of the fp64 reference on those priors, frame for frame.  -> frames redone in fp64"""
        import torch

        self.simulate("biawgn", param, codeword, seed, stream_id, frame0, B, max_iter, counters, flags=_lib.flag_prior_grid(k), hist_bins=hist_bins)
        cnt, frames = self.grid_violations()
        if cnt > len(frames):
            raise _lib.LdpcHipError("prior grid 2^-%d: %d frames beyond the exactness guard in one call -- the grid is too fine for this operating point" % (k, cnt))
        if cnt:
            h64 = self._fp64_sibling()
            pri = torch.cat([self.channel_device("biawgn", param, codeword, seed, stream_id, int(f), 1, prior_grid=k)[0] for f in frames])
            xh, it = h64.decode_device(pri.double().contiguous(), None, max_iter)
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.load().ldpc_count_errors(xh.data_ptr(), None, int(codeword), it.data_ptr(), len(frames), self.code.n, hist_bins,
                                                     counters.data_ptr(), st))
        return int(cnt)

    # ---- exact-in-fp32 rounds without a host round trip (montecarlo.DeviceSimulator sends them in blocks)
    REDO_ROWS = 512  # frames of one BLOCK of rounds the fp64 sibling re-decodes at most (a 65 536-frame round at 1-2 dB sets aside 1-15)

    def grid_list_dev(self):
        """Device address of this decoder's redo list ([0] = count, [1..] = global frame indices) -- ``ldpc_decoder_grid_list``."""
        import torch

        p, cap = ctypes.c_void_p(), ctypes.c_int64(0)
        _lib.check(_lib.load().ldpc_decoder_grid_list(self.h, ctypes.byref(p), ctypes.byref(cap), torch.cuda.current_stream().cuda_stream))
        return p.value

    def redo_on_stream(self, param, codeword, seed, stream_id, max_iter, k, counters, redone2, frame_base, round_stride, hist_bins=0):
        """Enqueue, on the CURRENT torch stream, the fp64 re-decode of every frame the guarded kernels set aside since the list was last reset
        (``simulate(..., flags=grid)``, one or several rounds): their priors are drawn again from the device-resident list
        (``ldpc_channel_list``), decoded by the fp64 sibling, counted into the row of their round -- ``counters`` is [rounds, >= 4 + hist_bins],
        round = (frame - frame_base) / round_stride -- and the list is reset.  ``redone2`` (int64[2], device) += [frames counted, 1 if the list
        was longer than REDO_ROWS].  No host synchronisation when the sibling runs on the LDS-resident kernels."""
        import torch

        lib, n, rows = _lib.load(), self.code.n, self.REDO_ROWS
        st = torch.cuda.current_stream().cuda_stream
        lst = self.grid_list_dev()
        if getattr(self, "_redo_pri", None) is None:
            self._redo_pri = torch.zeros((rows, n), dtype=torch.float32, device="cuda")
            self._redo_x = torch.zeros((rows, n), dtype=torch.uint8, device="cuda")
            self._redo_it = torch.zeros((rows,), dtype=torch.int32, device="cuda")
        h64 = self._fp64_sibling()
        _lib.check(lib.ldpc_channel_list(_lib.CHANNEL["biawgn"] | _lib.ch_prior_grid(k), _lib.DTYPE["f32"], float(param), int(codeword), int(seed),
                                         int(stream_id), lst, rows, n, self._redo_pri.data_ptr(), st))
        h64.decode_device(self._redo_pri.double(), None, max_iter, xhat=self._redo_x, iters=self._redo_it)
        nrounds, stride = (counters.shape[0], counters.stride(0)) if counters.dim() == 2 else (1, counters.shape[0])
        _lib.check(lib.ldpc_count_errors_list(self._redo_x.data_ptr(), int(codeword), self._redo_it.data_ptr(), lst, rows, n, hist_bins,
                                              counters.data_ptr(), int(stride), int(frame_base), int(round_stride), int(nrounds),
                                              redone2.data_ptr(), st))
        _lib.check(lib.ldpc_decoder_grid_list_reset(self.h, st))

    def last_repacks(self):
        """Streaming backend: how often the last decode re-formed its tiles from the live frames."""
        r = ctypes.c_int(0)
        _lib.check(_lib.load().ldpc_decoder_last_repacks(self.h, ctypes.byref(r)))
        return r.value

    def chunk_state(self):
        """Streaming backend: (frames per pass, how often a failed reservation halved it)."""
        c, r = ctypes.c_int64(0), ctypes.c_int(0)
        _lib.check(_lib.load().ldpc_decoder_chunk_state(self.h, ctypes.byref(c), ctypes.byref(r)))
        return c.value, r.value


class MlHandle:
    """Codebook of a short code resident in HBM + the exhaustive-search kernels (``ldpc_ml_*``)."""

    def __init__(self, codebook, channel, precision="f64", device=None):
        lib = _lib.load()
        self.cb = np.ascontiguousarray(codebook, dtype=np.uint8)
        self.K, self.n = self.cb.shape
        self.W = (self.K + 31) // 32
        self.channel, self.precision = channel, precision
        self.device = current_device() if device is None else device
        h = ctypes.c_void_p()
        _lib.check(lib.ldpc_ml_create(self.device, self.cb.ctypes.data, self.K, self.n, ctypes.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.load().ldpc_ml_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def obs_dtype(self):
        import torch

        if self.channel != "biawgn":
            return torch.uint8
        return torch.float64 if self.precision == "f64" else torch.float32

    def decode_device(self, y, coef, pick=None, want_mask=True):
        """y: contiguous CUDA tensor [B,n] (observations / symbols) -> dict of CUDA tensors: index, ties, best, xhat[, tie_mask]."""
        import torch

        if not y.is_cuda or not y.is_contiguous() or y.dtype != self.obs_dtype() or y.shape[1] != self.n:
            raise ValueError("y must be a contiguous CUDA tensor [B,%d] of dtype %s" % (self.n, self.obs_dtype()))
        B = y.shape[0]
        out = dict(index=torch.empty(B, dtype=torch.int32, device=y.device), ties=torch.empty(B, dtype=torch.int32, device=y.device),
                   best=torch.empty(B, dtype=torch.float64, device=y.device),
                   xhat=torch.empty((B, self.n), dtype=torch.uint8, device=y.device))
        if want_mask:
            out["tie_mask"] = torch.empty((B, self.W), dtype=torch.int32, device=y.device)
        if B == 0:
            return out
        c2 = (ctypes.c_double * 2)(float(coef[0]), float(coef[1]))
        st = torch.cuda.current_stream(y.device).cuda_stream
        _lib.check(_lib.load().ldpc_ml_decode(self.h, _lib.CHANNEL[self.channel], _lib.DTYPE[self.precision], c2, y.data_ptr(), B,
                                              None if pick is None else pick.data_ptr(), out["index"].data_ptr(),
                                              out["ties"].data_ptr(), out["tie_mask"].data_ptr() if want_mask else None,
                                              out["best"].data_ptr(), out["xhat"].data_ptr(), st))
        return out

    def simulate(self, channel, param, codeword, seed, stream_id, frame0, B, max_iter, counters, flags=0, hist_bins=0):
        """Same call shape as DecoderHandle.simulate (max_iter / flags / hist_bins have no meaning here)."""
        import torch

        st = torch.cuda.current_stream(counters.device).cuda_stream
        _lib.check(_lib.load().ldpc_ml_simulate(self.h, _lib.CHANNEL[channel], _lib.DTYPE[self.precision], float(param), int(codeword),
                                                int(seed), int(stream_id), int(frame0), int(B), counters.data_ptr(), st))


class AdmmHandle:
    """ADMM LP decoder workspace on one GPU (``ldpc_admm_*``)."""

    def __init__(self, code, device=None):
        lib = _lib.load()
        self.code_handle = code_handle(code, device)
        self.code = code
        h = ctypes.c_void_p()
        _lib.check(lib.ldpc_admm_create(self.code_handle.h, ctypes.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.load().ldpc_admm_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def decode_device(self, gamma, mu, eps, max_iter):
        """gamma: contiguous CUDA float64 [B,n] -> (x float64 [B,n] before pseudo_to_cw, iters int32 [B], converged uint8 [B])."""
        import torch

        if not gamma.is_cuda or gamma.dtype != torch.float64 or not gamma.is_contiguous() or gamma.shape[1] != self.code.n:
            raise ValueError("gamma must be a contiguous CUDA float64 tensor [B,%d]" % self.code.n)
        B = gamma.shape[0]
        x = torch.empty_like(gamma)
        iters = torch.empty(B, dtype=torch.int32, device=gamma.device)
        conv = torch.empty(B, dtype=torch.uint8, device=gamma.device)
        if B == 0:  # zero-size tensors have no storage to point at
            return x, iters, conv
        st = torch.cuda.current_stream(gamma.device).cuda_stream
        _lib.check(_lib.load().ldpc_admm_decode(self.h, gamma.data_ptr(), B, float(mu), float(eps), int(max_iter), x.data_ptr(),
                                                iters.data_ptr(), conv.data_ptr(), st))
        return x, iters, conv

    # knobs of the simulate() composition below, set by admm.ADMM
    mu, eps, allow_pseudo, on_iters = 3.0, 1e-5, False, None

    def last_repacks(self):
        """How often the last decode re-formed its tiles from the live frames."""
        r = ctypes.c_int(0)
        _lib.check(_lib.load().ldpc_admm_last_repacks(self.h, ctypes.byref(r)))
        return r.value

    def last_backend(self):
        """Kernels of the last decode: "stream" (state in HBM) or "lds" (one workgroup per frame, state in the LDS)."""
        r = ctypes.c_int(0)
        _lib.check(_lib.load().ldpc_admm_last_backend(self.h, ctypes.byref(r)))
        return "lds" if r.value == 1 else "stream"

    def simulate(self, channel, param, codeword, seed, stream_id, frame0, B, max_iter, counters, flags=0, hist_bins=0):
        """Same call shape as DecoderHandle.simulate: device channel kernel -> LLRs -> ADMM -> pseudo_to_cw -> counters, all
        on the GPU (a composition of ldpc_channel / ldpc_admm_decode / torch element-wise ops; no host noise)."""
        import torch

        if B <= 0:
            return
        lib, n = _lib.load(), self.code.n
        st = torch.cuda.current_stream(counters.device).cuda_stream
        step = 1 << 15
        for b0 in range(0, int(B), step):
            nb = min(step, int(B) - b0)
            if channel == "bec":
                y = torch.empty((nb, n), dtype=torch.uint8, device=counters.device)
                _lib.check(lib.ldpc_channel(_lib.CHANNEL[channel], 0, float(param), int(codeword), int(seed), int(stream_id), int(frame0) + b0,
                                            nb, n, None, y.data_ptr(), st))
                gamma = torch.tensor([1e8, -1e8, 0.0], dtype=torch.float64, device=counters.device)[y.long()]  # src/bec.py:41
            else:
                gamma = torch.empty((nb, n), dtype=torch.float64, device=counters.device)
                y = torch.empty((nb, n), dtype=torch.uint8, device=counters.device) if channel == "bsc" else None
                _lib.check(lib.ldpc_channel(_lib.CHANNEL[channel], _lib.DTYPE["f64"], float(param), int(codeword), int(seed), int(stream_id),
                                            int(frame0) + b0, nb, n, gamma.data_ptr(), None if y is None else y.data_ptr(), st))
            x, iters, _ = self.decode_device(gamma.contiguous(), self.mu, self.eps, max_iter)
            if self.allow_pseudo:  # src/math_utils.py:28-34, then `x != x_hat` as src/main.py:41
                x = torch.where(x < 1e-8, torch.zeros_like(x), x)
                x = torch.where(1 - x < 1e-8, torch.ones_like(x), x)
                wrong = x != float(codeword)
            else:
                wrong = (x > .5) != bool(codeword)
            err = wrong.sum(dim=1)
            counters[_lib.CNT_TOT] += nb
            counters[_lib.CNT_WEC] += (err > 0).sum()
            counters[_lib.CNT_BEC] += err.sum()
            counters[_lib.CNT_ITER_SUM] += iters.sum()
            if hist_bins:
                counters[_lib.CNT_HIST0:_lib.CNT_HIST0 + hist_bins] += torch.bincount(iters.clamp(max=hist_bins - 1).long(), minlength=hist_bins)
            if self.on_iters is not None:
                self.on_iters(iters)
