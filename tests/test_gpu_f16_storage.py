"""fp16 STORAGE mode of the streaming kernels (LDPC_DTYPE_F16; SURVEY 8(b) dtype list, 8(d) 31 200 B row): check messages kept as fp16,
arithmetic and marginals fp32.  A throughput mode held to a STATED TOLERANCE, never the parity mode:

  * per sweep, against the fp32 streaming kernels AND against the oracle on identical priors (no early exit, sweeps 1..3): every marginal within
    TOL * (1 + |fp32 marginal|), TOL = 1e-2 (measured maxima 2e-3 .. 5e-3) -- each stored message is rounded to 11 significant bits (relative 2^-11 = 4.9e-4), a
    marginal sums up to dv of them, and three sweeps compound it;
  * decisions of the first sweep identical wherever the fp32 marginal is not within the tolerance of zero;
  * word / bit error rates at the reference's published operating points within 4 sigma / the spread of the published files
    (tests/golden/published_curves.json: biawgn-1200_3_6_rand_ldpc_1-{MSA-10-1, SPA-10-0}).
"""
import numpy as np
import pytest

import bp_oracle as O
from helpers import golden_edges
from test_gpu_large_codes import _published_point

pytestmark = pytest.mark.gpu
TOL = 1e-2


def _code(name="1200_3_6_rand_ldpc_1"):
    from ldpc_decoders_amd.codes import Code

    g = golden_edges(name)
    return g, Code.from_edges(g.m, g.n, g.chk, g.var)


@pytest.mark.parametrize("alg", ["MSA", "SPA"])
@pytest.mark.parametrize("name,B", [("1200_3_6_rand_ldpc_1", 300), ("1200_rho_x5_rand_ldpc_5", 130)])  # 300 = 2 pair-tiles + a ragged third; irregular degrees
def test_per_sweep_operator_tolerance_against_the_fp32_kernels(alg, name, B):
    import torch
    from ldpc_decoders_amd import _lib
    from ldpc_decoders_amd._device import DecoderHandle

    g, code = _code(name)
    rng = np.random.RandomState(17)
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(2.0)), (B, g.n))
    pri = torch.from_numpy(O.biawgn_priors(y, 2.0).astype(np.float32)).cuda()
    h16, h32 = DecoderHandle(code, alg, "f16"), DecoderHandle(code, alg, "f32", "stream")
    for sweeps in (1, 2, 3):
        x16, i16, m16 = h16.decode_soft_device(pri, None, sweeps, flags=_lib.FLAG_NO_EARLY_EXIT)
        x32, i32, m32 = h32.decode_soft_device(pri, None, sweeps, flags=_lib.FLAG_NO_EARLY_EXIT)
        assert h16.last_stats()[0] == "stream" and (i16 == sweeps).all() and (i32 == sweeps).all()
        m16, m32 = m16.cpu().numpy().astype(np.float64), m32.cpu().numpy().astype(np.float64)
        fin = np.isfinite(m32)
        err = np.abs(m16 - m32)[fin] / (1 + np.abs(m32[fin]))
        print("%s %s, %d sweep(s): max |marg16 - marg32| / (1 + |marg32|) = %.2e over %d marginals" % (name, alg, sweeps, err.max(), err.size))
        assert err.max() <= TOL and (np.isfinite(m16) == fin).all()
        if sweeps == 1:
            clear = np.abs(m32) > TOL * (1 + np.abs(m32))
            assert ((x16.cpu().numpy() == x32.cpu().numpy()) | ~clear).all()


@pytest.mark.parametrize("alg", ["MSA", "SPA"])
def test_per_sweep_operator_tolerance_against_the_oracle(alg):
    # the same bound against the ORACLE (oracle/bp_oracle.py, fp64, the reference's own formulas: src/bpa.py:31-38): marginals of the frames
    # still running after sweeps 1..3, on the fp32 priors the device is given
    import torch
    from ldpc_decoders_amd import _lib
    from ldpc_decoders_amd._device import DecoderHandle

    g, code = _code()
    B = 96
    rng = np.random.RandomState(23)
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(2.0)), (B, g.n))
    pri32 = O.biawgn_priors(y, 2.0).astype(np.float32)
    _, _, trace = O.bp_decode(g, alg, (pri32 < 0).astype(np.float64), pri32.astype(np.float64), 3, return_trace=True)
    h16 = DecoderHandle(code, alg, "f16")
    pri = torch.from_numpy(pri32).cuda()
    for sweeps in (1, 2, 3):
        _, _, m16 = h16.decode_soft_device(pri, None, sweeps, flags=_lib.FLAG_NO_EARLY_EXIT)
        m16, ref = m16.cpu().numpy().astype(np.float64), trace[sweeps - 1]
        running = ~np.isnan(ref).all(axis=1)          # frames the oracle was still decoding in this sweep
        ok = np.isfinite(ref) & running[:, None]
        err = np.abs(m16 - ref)[ok] / (1 + np.abs(ref[ok]))
        print("%s, %d sweep(s): max |marg16 - oracle| / (1 + |oracle|) = %.2e over %d marginals of %d frames" % (alg, sweeps, err.max(), err.size, running.sum()))
        assert running.sum() >= B // 2 and err.max() <= TOL


@pytest.mark.parametrize("alg,max_iter,cw,snr", [("MSA", 10, 1, "2.0"), ("MSA", 10, 1, "2.25"), ("SPA", 10, 0, "2.0")])
def test_error_rates_at_the_published_operating_points(alg, max_iter, cw, snr):
    import torch
    from ldpc_decoders_amd._device import DecoderHandle

    g, code = _code()
    ref = _published_point("biawgn", "1200_3_6_rand_ldpc_1", alg, max_iter, cw, snr)
    h = DecoderHandle(code, alg, "f16")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    h.simulate("biawgn", float(snr), cw, 21, 4, 0, 8192, max_iter, cnt)
    assert h.last_stats()[0] == "stream"
    tot, wec, bec = (int(v) for v in cnt.cpu().numpy()[:3])
    wer, ber = wec / tot, bec / (tot * code.n)
    sigma = np.sqrt(ref["wer"] * (1 - ref["wer"]) * (1 / ref["tot"] + 1 / tot))
    print("fp16 storage %s %s dB max_iter=%d: WER %.4f (published %.4f +- %.4f), BER %.3e (published %.3e)" % (alg, snr, max_iter, wer, ref["wer"], sigma, ber, ref["ber"]))
    assert tot == 8192 and abs(wer - ref["wer"]) <= 4 * sigma
    assert 0.6 <= (ber / wer) / (ref["ber"] / ref["wer"]) <= 1.6


def test_fp16_mode_statistics_equal_the_fp32_mode_on_identical_noise():
    # same Philox frames through both modes at 2.0 dB, 50 sweeps: word-error counts within the binomial spread of their difference
    import torch
    from ldpc_decoders_amd._device import DecoderHandle

    g, code = _code()
    res = {}
    for prec in ("f16", "f32"):
        h = DecoderHandle(code, "MSA", prec, "stream")
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        h.simulate("biawgn", 2.0, 0, 5, 1, 0, 16384, 50, cnt)
        res[prec] = cnt.cpu().numpy()
    a, b = res["f16"], res["f32"]
    print("2.0 dB, 16 384 identical frames: fp16-storage wec %d, mean sweeps %.2f; fp32 wec %d, mean sweeps %.2f" % (a[1], a[3] / a[0], b[1], b[3] / b[0]))
    assert a[0] == b[0] == 16384 and abs(int(a[1]) - int(b[1])) <= 4 * np.sqrt(b[1]) + 8
    assert abs(a[3] / a[0] - b[3] / b[0]) <= 0.5


def test_config5_size_properties_of_the_fp16_mode():
    # BASELINE config 5 ((3,6), n = 64 800) in the fp16 storage mode, on priors both modes are given: every frame that leaves early carries a
    # codeword (size-independent), the fp16 mode decodes what the fp32 kernels decode on all but a stated fraction of the frames (tolerance
    # mode: a frame near its threshold may need a few sweeps more or fewer), and with the frame repack on pair-tiles active
    import torch
    from ldpc_decoders_amd import codes
    from ldpc_decoders_amd._device import DecoderHandle

    code = codes.rand_reg_ldpc(64800, 3, 6, np.random.RandomState(8))
    rng = np.random.RandomState(3)
    B = 640
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(1.8)), (B, code.n))   # just above this code's min-sum threshold: frames leave after 20..50 sweeps
    pri = torch.from_numpy(O.biawgn_priors(y, 1.8).astype(np.float32)).cuda()
    h16, h32 = DecoderHandle(code, "MSA", "f16"), DecoderHandle(code, "MSA", "f32", "stream")
    x16, i16 = h16.decode_device(pri, None, 50)
    x32, i32 = h32.decode_device(pri, None, 50)
    x16, i16, x32, i32 = x16.cpu().numpy(), i16.cpu().numpy(), x32.cpu().numpy(), i32.cpu().numpy()
    done16, done32 = i16 < 50, i32 < 50
    print("config 5, fp16 storage: %d of %d frames leave early (fp32: %d); iteration counts differ on %d frames, by at most %d; repacks %d" % (
        done16.sum(), B, done32.sum(), (i16 != i32).sum(), np.abs(i16 - i32).max(), h16.last_repacks()))
    assert done16.any() and code.syndrome(x16[done16]).sum() == 0
    assert (done16 != done32).mean() <= 0.02            # the same frames decode
    both = done16 & done32
    assert (x16[both] == x32[both]).all()                # ... to the same words (the all-zero word here, but through different messages)
    assert np.abs(i16[both] - i32[both]).max() <= 5      # measured: 92 of 640 iteration counts differ, by at most 5; 618 / 620 frames leave early, 4 repacks
    assert h16.last_repacks() >= 1
