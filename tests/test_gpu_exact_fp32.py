"""Exact-in-fp32 min-sum ("alternative exact-in-fp32 trick"): priors rounded to multiples of 2^-k make every sum of the
min-sum recursion exactly representable in fp32 for as long as the messages stay small, so the fp32 LDS kernels reproduce the FP64
reference -- `bpa.MSA.decode(y, priors)` fed the same priors (src/bpa.py:17) -- bit for bit, and an in-kernel guard reports any frame in
which a message left the exact range.  Oracle: the fp64 C restatement on the very priors the device produced."""
import numpy as np
import pytest

import bp_oracle as O
import c_oracle as C
from helpers import golden_edges

pytestmark = pytest.mark.gpu
K = 8  # grid 2^-8: what SURVEY probed (|marginal| < 100 over 50 sweeps: far below the guard's 2^13)


def _decode_against_the_fp64_oracle(code, g, snr, B, max_iter=50):
    import torch
    from ldpc_decoders_amd import _lib
    from ldpc_decoders_amd._device import DecoderHandle

    h = DecoderHandle(code, "MSA", "f32", "fused")
    pri, _ = h.channel_device("biawgn", snr, 0, 0x5EED1200, 3, 0, B, prior_grid=K)
    p = pri.cpu().numpy()
    assert p.dtype == np.float32 and (p * 2.0 ** K == np.rint(p * 2.0 ** K)).all() and len(np.unique(p[0])) > 100
    # the same noise without the grid differs only by the rounding
    raw, _ = h.channel_device("biawgn", snr, 0, 0x5EED1200, 3, 0, 256)
    assert np.abs(raw.cpu().numpy() - p[:256]).max() <= 2.0 ** -(K + 1) * 1.0001
    xo, io = C.bp_decode(g, "MSA", None, p.astype(np.float64), max_iter, dtype=np.float64)  # the reference's arithmetic on the same priors
    # the raw kernel: frames the guard vouches for are the fp64 reference's, bit for bit; the others are MARKED (iters < 0) -- min-sum
    # messages of a frame caught in a trapping set grow geometrically and leave the range where fp32 sums are exact
    xr, ir = h.decode_device(pri, None, max_iter, flags=_lib.flag_prior_grid(K))
    assert h.last_stats()[0] == "fused"
    xr, ir = xr.cpu().numpy(), ir.cpu().numpy()
    ok = ir >= 0
    nviol, listed = h.grid_violations()
    assert nviol == (~ok).sum() == len(listed) and (~ok).sum() <= max(4, B // 2000), "%d frames beyond the guard" % (~ok).sum()
    assert sorted(listed) == sorted(np.flatnonzero(~ok))
    assert (xr[ok] == xo[ok]).all() and (ir[ok] == io[ok]).all(), "a frame the guard vouched for differs from fp64"
    print("exact-in-fp32: %d of %d frames beyond the guard (sweeps there: %s)" % ((~ok).sum(), B, (-1 - ir[~ok]).tolist()))
    # the mode as the host uses it: marked frames decoded again in fp64 -> EVERY frame is the fp64 reference's
    xh, it, redone = h.decode_device_exact_fp32(pri, max_iter, K)
    xh, it = xh.cpu().numpy(), it.cpu().numpy()
    assert redone == (~ok).sum() and (xh == xo).all() and (it == io).all()
    # channel + decode + count in ONE kernel draws the same quantised priors: with the set-aside frames redone, its counters are those of
    # the fp64 oracle
    cnt = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
    assert h.simulate_exact_fp32(snr, 0, 0x5EED1200, 3, 0, B, max_iter, cnt, K, hist_bins=51) == redone
    c = cnt.cpu().numpy()
    err = (xo != 0).sum(axis=1)
    assert (c[0], c[1], c[2], c[3]) == (B, (err > 0).sum(), err.sum(), io.sum())
    assert (c[4:] == np.bincount(np.minimum(io, 50), minlength=51)).all()
    return h, pri, (it < max_iter).mean()


def test_config2_fp32_on_a_prior_grid_equals_the_fp64_reference_on_every_frame():
    from ldpc_decoders_amd import _lib
    from ldpc_decoders_amd.codes import Code

    g = golden_edges("1200_3_6_rand_ldpc_1")
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    h, pri, conv = _decode_against_the_fp64_oracle(code, g, 2.0, 16384)
    assert 0.5 < conv < 1.0  # converging and non-converging (chaotic) frames alike
    # WITHOUT the grid the same kernel family is not frame-exact on real-valued priors: the mode is what buys it
    # -- and with a grid so fine that sums leave the 24-bit range the guard fires
    xh, it = h.decode_device(pri, None, 50, flags=_lib.flag_prior_grid(20))  # exact only below 2^(24-20-3) = 2: nearly every frame trips
    assert h.grid_violations()[0] > 8000 and (it < 0).sum().item() > 8000


def test_config4_fp32_on_a_prior_grid_equals_the_fp64_reference_on_every_frame():
    from ldpc_decoders_amd import codes

    code = codes.rand_irregular_ldpc(10000, codes.LAMBDA_RHO_X5_HALF_RATE, 6, np.random.RandomState(4))
    g = O.Edges(code.m, code.n, code.edge_chk, code.edge_var)
    _, _, conv = _decode_against_the_fp64_oracle(code, g, 1.8, 16384)
    assert conv > 0.5


def test_prior_grid_where_no_guarded_kernel_exists_is_refused_not_ignored():
    import torch
    from ldpc_decoders_amd import _lib
    from ldpc_decoders_amd.codes import Code
    from ldpc_decoders_amd._device import DecoderHandle

    g = golden_edges("1200_3_6_rand_ldpc_1")
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    h = DecoderHandle(code, "MSA", "f32", "stream")
    with pytest.raises(_lib.LdpcHipError, match="prior grid"):
        h.simulate("biawgn", 2.0, 0, 1, 0, 0, 256, 50, cnt, flags=_lib.flag_prior_grid(K))
    # fp64 decoders take grid priors through the composed path (quantised by the channel kernel; fp64 sums need no guard)
    h64 = DecoderHandle(code, "MSA", "f64")
    h64.simulate("biawgn", 2.0, 0, 1, 0, 0, 256, 50, cnt, flags=_lib.flag_prior_grid(K))
    h32 = DecoderHandle(code, "MSA", "f32")
    cnt32 = torch.zeros(4, dtype=torch.int64, device="cuda")
    h32.simulate("biawgn", 2.0, 0, 1, 0, 0, 256, 50, cnt32, flags=_lib.flag_prior_grid(K))
    # fp32 priors of the in-kernel channel and fp64 priors of the channel kernel round to the same grid point except at ties of the
    # rounding: the two runs agree on the frame count and are both exact for their own priors
    assert int(cnt[0]) == 256 and int(cnt32[0]) + h32.grid_violations()[0] == 256


def test_cli_prior_grid_counts_what_the_fp64_reference_counts(tmp_path):
    # `main.py biawgn <code> MSA --prior-grid 8`: the counters of the run are those of the fp64 oracle on the priors the device drew
    from ldpc_decoders_amd import main as M
    from ldpc_decoders_amd.codes import Code
    from ldpc_decoders_amd._device import DecoderHandle

    res = M.main(["biawgn", "1200_3_6_rand_ldpc_1", "MSA", "--codeword", "0", "--min-wec", "300", "--max-iter", "50", "--params", "2.0",
                  "--batch", "4096", "--prior-grid", str(K), "--data_dir", str(tmp_path), "--console"])[2.0]
    assert res["tot"] % 4096 == 0 and res["wec"] >= 300
    g = golden_edges("1200_3_6_rand_ldpc_1")
    h = DecoderHandle(Code.from_edges(g.m, g.n, g.chk, g.var), "MSA", "f32")
    pri, _ = h.channel_device("biawgn", 2.0, 0, 0x5EED1200, 0, 0, res["tot"], prior_grid=K)  # stream 0 = first --params value
    xo, _ = C.bp_decode(g, "MSA", None, pri.cpu().numpy().astype(np.float64), 50, dtype=np.float64)
    err = (xo != 0).sum(axis=1)
    assert (res["wec"], res["bec"]) == ((err > 0).sum(), err.sum())
