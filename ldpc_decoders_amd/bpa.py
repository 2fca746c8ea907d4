"""Flooding belief propagation in the LLR domain on the GPU -- mirror of the reference's ``src/bpa.py``.

Same classes and call shapes: ``SPA(parity_mtx, max_iter=..)`` / ``MSA(parity_mtx, max_iter=..)`` (extra keyword
arguments are ignored, the reference splats all CLI flags into the constructor, src/main.py:26), ``id_keys``, and
``decode(y, priors) -> x_hat`` (src/bpa.py:17-63).  Added: ``decode_batch`` (many frames per call, numpy or CUDA
tensors).  The arithmetic is in libldpc_hip.so; this file only moves buffers.

Extra constructor keywords (all optional):
  precision  'f64' (default here: bit-exact min-sum against the reference) or 'f32' (throughput mode)
  backend    'auto' | 'stream' | 'fused'
  device     GPU ordinal (default: torch's current device, else 0)
"""
import numpy as np

from ._device import DecoderHandle, as_code


class BPA:
    id_keys = ["max_iter"]
    alg = None

    def __init__(self, parity_mtx, **kwargs):
        self.max_iter = kwargs["max_iter"]  # KeyError if absent, as upstream (src/bpa.py:10)
        self.code = as_code(parity_mtx)
        self.precision = kwargs.get("precision") or "f64"
        self.handle = DecoderHandle(self.code, self.alg, self.precision, kwargs.get("backend") or "auto", kwargs.get("device"))
        self.last_iters = None

    @property
    def parity_mtx(self):
        return self.code.parity_mtx

    def _iter0_word(self, y):
        """Hard word for the iteration-0 syndrome check (src/bpa.py:20,29), or None when it cannot pass.

        Upstream tests ``(H @ y) % 2 == 0`` on the RAW received vector: meaningful for the BSC (y in {0,1}); for a
        real-valued BI-AWGN observation it passes only if every check sum is an even integer, which requires an
        integer-valued y -- handled here on the host only in that (measure-zero) case."""
        y = np.asarray(y)
        if y.dtype.kind == "f" and y.size and y.flat[0] != np.floor(y.flat[0]):
            return None, None  # a real-valued observation (every BI-AWGN frame): decided on the first entry, not on a pass over the frame
        if y.dtype.kind in "biu" or np.all(y == np.floor(y)):
            yi = np.asarray(y, dtype=np.int64)
            if ((yi == 0) | (yi == 1)).all():
                return yi.astype(np.uint8), None
            ok = (self.code.syndrome(yi) == 0).all(axis=-1)  # integer-valued, not binary: decide on the host
            return None, np.atleast_1d(ok)
        return None, None

    def decode(self, y, priors):
        y = np.asarray(y)
        y0, host_ok = self._iter0_word(y)
        if host_ok is not None and host_ok[0]:  # passes the iteration-0 test of src/bpa.py:28-29 for every max_iter (0 included)
            self.last_iters = np.zeros(1, dtype=np.int32)
            return y
        xhat, iters = self.handle.decode_host(np.asarray(priors), y0, self.max_iter)
        self.last_iters = iters
        if iters[0] == 0 and y0 is not None:
            return y  # left at the iteration-0 check: upstream returns the received object itself
        return xhat[0].astype(np.int64)

    def decode_batch(self, y, priors):
        """[B,n] frames -> (x_hat uint8 [B,n], iters int32 [B]).  numpy in -> numpy out; CUDA tensors in -> CUDA
        tensors out (no copies; y may be None or a uint8 tensor of hard received words)."""
        if hasattr(priors, "is_cuda"):
            out = self.handle.decode_device(priors, y, self.max_iter)
            self.last_iters = out[1]
            return out
        y0 = None
        if y is not None:
            y0, host_ok = self._iter0_word(y)
            if host_ok is not None and host_ok.any():
                raise ValueError("integer-valued non-binary received words are only supported one frame at a time")
        out = self.handle.decode_host(priors, y0, self.max_iter)
        self.last_iters = out[1]
        return out


class SPA(BPA):
    """Sum-product: tanh-product check rule (src/bpa.py:66-75)."""
    alg = "SPA"


class MSA(BPA):
    """Min-sum: two-min + sign-parity check rule (src/bpa.py:78-102)."""
    alg = "MSA"
