"""Packed decisions at the C ABI (`uint32* xhat_bits [B, ceil(n/32)]`): ldpc_decode_bits, ldpc_decode_host_bits and
ldpc_count_errors_bits must carry exactly the information of the byte form -- BPA.decode's n hard decisions (src/bpa.py:62) / the erasure
decoder's {0,1,2} symbols (src/bec.py:120) -- on every backend: LDS-resident kernels (bytes packed by a kernel), streaming kernels (words
written straight from the decision bit planes, with and without the frame repack, fp16 storage mode), erasure decoder (second mask)."""
import numpy as np
import pytest

import bp_oracle as O
import c_oracle as C
from helpers import golden_edges

pytestmark = pytest.mark.gpu


def _code(name):
    from ldpc_decoders_amd.codes import Code

    g = golden_edges(name)
    return g, Code.from_edges(g.m, g.n, g.chk, g.var)


def _priors(seed, B, n, snr, dt):
    rng = np.random.RandomState(seed)
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(snr)), (B, n))
    return O.biawgn_priors(y, snr).astype(dt)


@pytest.mark.parametrize("name,prec,backend,B,snr", [
    ("1200_3_6_rand_ldpc_1", "f64", "auto", 3001, 2.0),      # LDS-resident fp64 kernel, ragged batch
    ("1200_3_6_rand_ldpc_1", "f32", "stream", 64 * 40 + 17, 2.2),  # streaming kernels: frames leave one by one, the repack fires
    ("1200_3_6_rand_ldpc_1", "f16", "stream", 64 * 41 + 17, 2.3),  # fp16 storage mode (pair tiles)
    ("1200_rho_x5_rand_ldpc_5", "f32", "stream", 777, 1.8),  # irregular degrees, m = 599
    ("7_4_hamming", "f64", "stream", 130, 3.0),              # n = 7: one partial word per frame
    ("7_4_hamming", "f64", "auto", 130, 3.0),
    ("4_2_test", "f32", "stream", 65, 1.0),
])
def test_packed_decisions_equal_the_byte_form(monkeypatch, name, prec, backend, B, snr):
    import torch
    from ldpc_decoders_amd._device import DecoderHandle, unpack_bits

    g, code = _code(name)
    dt = np.float64 if prec == "f64" else np.float32
    pri = torch.from_numpy(_priors(3, B, g.n, snr, dt)).cuda()
    if backend == "stream":
        monkeypatch.setenv("LDPC_STREAM_REPACK_FILL", "0.97")
    h = DecoderHandle(code, "MSA", prec, backend)
    xb, ib = h.decode_device(pri, None, 50)
    bits, era, it = h.decode_device_bits(pri, None, 50)
    assert era is None and bits.shape == (B, (g.n + 31) // 32)
    if backend == "stream" and g.n == 1200:
        assert h.last_stats()[0] == "stream" and h.last_repacks() >= 1
    x = unpack_bits(bits.cpu().numpy(), g.n)
    assert (x == xb.cpu().numpy()).all() and (it == ib).all()
    # padding bits of the last word are zero
    W = (g.n + 31) // 32
    if g.n % 32:
        assert (bits.cpu().numpy().view(np.uint32)[:, W - 1] >> (g.n % 32) == 0).all()
    if prec != "f16":  # min-sum is exact arithmetic: the C oracle in the same precision gives the same bits
        xo, io = C.bp_decode(g, "MSA", None, pri.cpu().numpy(), 50, dtype=dt)
        assert (x == xo).all() and (it.cpu().numpy() == io).all()
    # counters from the packed form == counters from the bytes, for both sent words
    from ldpc_decoders_amd import _lib

    for cw in (0, 1):
        c_bytes = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
        c_bits = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().ldpc_count_errors(xb.data_ptr(), None, cw, ib.data_ptr(), B, g.n, 51, c_bytes.data_ptr(), st))
        h.count_errors_bits(bits, None, it, c_bits, codeword=cw, hist_bins=51)
        assert (c_bytes == c_bits).all() and int(c_bits[0]) == B
    # host entry points: bytes (bits over PCIe, expanded on the host) and packed
    xh, ih = h.decode_host(pri.cpu().numpy(), None, 50)
    assert (xh == x).all() and (ih == ib.cpu().numpy()).all()
    hb, he, hi = h.decode_host_bits(pri.cpu().numpy(), None, 50)
    assert he is None and (hb == bits.cpu().numpy().view(np.uint32)).all() and (hi == ih).all()


@pytest.mark.parametrize("name,backend,B", [("1200_3_6_rand_ldpc_1", "auto", 2049), ("1200_3_6_rand_ldpc_1", "stream", 2049), ("7_4_hamming", "auto", 100)])
def test_packed_decisions_of_the_erasure_decoder(name, backend, B):
    import torch
    from ldpc_decoders_amd import _lib
    from ldpc_decoders_amd._device import DecoderHandle, unpack_bits

    g, code = _code(name)
    rng = np.random.RandomState(8)
    word = np.zeros(g.n, dtype=np.int64)
    y = O.bec_send(np.broadcast_to(word, (B, g.n)), 0.42, rng).astype(np.uint8)  # around the threshold: resolved frames and stopping sets
    yd = torch.from_numpy(y).cuda()
    h = DecoderHandle(code, "BEC", "f32", backend)
    xb, ib = h.decode_device(None, yd, 50)
    bits, era, it = h.decode_device_bits(None, yd, 50)
    x = unpack_bits(bits.cpu().numpy(), g.n, era.cpu().numpy())
    assert (x == xb.cpu().numpy()).all() and (it == ib).all()
    xo, io = C.bec_decode(g, y, 50)
    assert (x == xo).all() and (it.cpu().numpy() == io).all()
    assert (x == 2).any() and (x == 0).all(axis=1).any()
    assert ((bits.cpu().numpy() & era.cpu().numpy()) == 0).all()  # an erased symbol has decision bit 0
    c_bytes = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
    c_bits = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.load().ldpc_count_errors(xb.data_ptr(), None, 0, ib.data_ptr(), B, g.n, 51, c_bytes.data_ptr(), st))
    h.count_errors_bits(bits, era, it, c_bits, codeword=0, hist_bins=51)
    assert (c_bytes == c_bits).all()
    xh, ih = h.decode_host(None, y, 50)
    assert (xh == xo).all() and (ih == io).all()
    hb, he, hi = h.decode_host_bits(None, y, 50)
    assert (unpack_bits(hb, g.n, he) == xo).all() and (hi == io).all()
    with pytest.raises(_lib.LdpcHipError):  # the erasure decoder cannot drop its "still erased" mask
        _lib.check(_lib.load().ldpc_decode_bits(h.h, None, yd.data_ptr(), B, 50, 0, bits.data_ptr(), None, it.data_ptr(), st))


def test_prior_grid_guard_is_refused_on_the_streaming_kernels():
    # the exactness guard lives in the LDS-resident fp32 min-sum kernels; a decoder that runs on the streaming
    # kernels must refuse LDPC_FLAG_PRIOR_GRID instead of returning frames nobody vouched for (ldpc_simulate already did)
    import torch
    from ldpc_decoders_amd import _lib
    from ldpc_decoders_amd._device import DecoderHandle

    g, code = _code("1200_3_6_rand_ldpc_1")
    h32 = DecoderHandle(code, "MSA", "f32", "stream")
    pri, _ = h32.channel_device("biawgn", 2.0, 0, 1, 0, 0, 256, prior_grid=8)
    with pytest.raises(_lib.LdpcHipError, match="guard"):
        h32.decode_device_exact_fp32(pri, 50, 8)
    with pytest.raises(_lib.LdpcHipError, match="guard"):
        h32.decode_device(pri, None, 50, flags=_lib.flag_prior_grid(8))
    marg = torch.zeros_like(pri)
    xh = torch.empty((256, g.n), dtype=torch.uint8, device="cuda")
    it = torch.empty(256, dtype=torch.int32, device="cuda")
    rc = _lib.load().ldpc_decode_soft(h32.h, pri.data_ptr(), None, 256, 50, _lib.flag_prior_grid(8), xh.data_ptr(), it.data_ptr(), marg.data_ptr(), None)
    assert rc == -4  # LDPC_E_UNSUPPORTED
    h16 = DecoderHandle(code, "MSA", "f16", "stream")
    with pytest.raises(_lib.LdpcHipError, match="guard"):
        h16.decode_device(pri, None, 50, flags=_lib.flag_prior_grid(8))
    # the LDS-resident fp32 kernel accepts it; an fp64 decoder needs no guard (the flag is a no-op), on either backend
    hf = DecoderHandle(code, "MSA", "f32", "auto")
    x1, i1, redone = hf.decode_device_exact_fp32(pri, 50, 8)
    for backend in ("auto", "stream"):
        h64 = DecoderHandle(code, "MSA", "f64", backend)
        x2, i2 = h64.decode_device(pri.double().contiguous(), None, 50, flags=_lib.flag_prior_grid(8))
        assert (x1 == x2).all() and (i1 == i2).all()


def test_random_codeword_simulation_in_the_fp16_storage_mode():
    # `--precision f16 --codeword -1` reached ldpc_channel_words with the decoder dtype (2) instead of the io dtype
    import torch
    from ldpc_decoders_amd import codes
    from ldpc_decoders_amd._device import DecoderHandle

    code = codes.get_code("7_4_hamming")
    cnt16 = torch.zeros(4, dtype=torch.int64, device="cuda")
    cnt32 = torch.zeros(4, dtype=torch.int64, device="cuda")
    DecoderHandle(code, "MSA", "f16", "stream").simulate("biawgn", 3.0, -1, 5, 0, 0, 4096, 10, cnt16)
    DecoderHandle(code, "MSA", "f32", "stream").simulate("biawgn", 3.0, -1, 5, 0, 0, 4096, 10, cnt32)
    assert int(cnt16[0]) == 4096 and int(cnt32[0]) == 4096
    assert abs(int(cnt16[1]) - int(cnt32[1])) <= 40  # same frames, same noise: fp16 storage moves only a few borderline frames


def test_chunk_halving_retry_after_a_failed_reservation(monkeypatch):
    """ldpc_decode on the streaming kernels: a workspace reservation that fails (other allocations took the memory the chunk was sized
    from) halves the chunk, gives back what the failed attempt had reserved, and decodes the SAME result.  The failure is injected
    (LDPC_TEST_FAIL_RESERVE: the K-th growing reservation asks for 2^50 bytes -- a genuine hipMalloc error, sticky status included,
    which ROCm 7 would hand to the next hipGetLastError())."""
    import torch

    from ldpc_decoders_amd import codes
    from ldpc_decoders_amd._device import DecoderHandle

    code = codes.get_code("1200_3_6_rand_ldpc_1")
    rng = np.random.RandomState(5)
    B = 1024
    pri = torch.from_numpy((2.0 + 1.3 * rng.standard_normal((B, code.n))).astype(np.float32)).cuda()
    ref = DecoderHandle(code, "MSA", "f32", "stream")
    x0, it0 = ref.decode_device(pri, None, 30)
    assert ref.chunk_state()[1] == 0
    for k in (1, 2, 4):   # the failure lands on different workspaces of the first pass
        dec = DecoderHandle(code, "MSA", "f32", "stream")
        monkeypatch.setenv("LDPC_TEST_FAIL_RESERVE", str(k))
        x1, it1 = dec.decode_device(pri, None, 30)
        monkeypatch.delenv("LDPC_TEST_FAIL_RESERVE")
        torch.cuda.synchronize()
        chunk, retries = dec.chunk_state()
        assert retries == 1 and chunk == B // 2, (k, chunk, retries)
        assert torch.equal(x0, x1) and torch.equal(it0, it1), k
        x2, it2 = dec.decode_device(pri, None, 30)   # the decoder stays usable at the smaller chunk
        assert torch.equal(x0, x2) and dec.chunk_state() == (B // 2, 1)
