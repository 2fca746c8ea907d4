"""GPU parity tests of the LDS-resident (fused) kernels beyond check degree 6: the reference's generators take any (l, r)
(src/codes.py:108-120,165-171) and any rho (src/ldpc.py:149-155, check degree rho + 1).  Generated (3,4)-, (4,8)-, (3,5)-regular and
rho = x^4 / x^6 irregular codes, fp32 and fp64, against the C oracle (min-sum, erasure decoder: bit-exact) and the streaming kernels."""
import numpy as np
import pytest

import bp_oracle as O
import c_oracle as C

pytestmark = pytest.mark.gpu

CODES = {
    "reg_3_4": lambda codes: codes.rand_reg_ldpc(1200, 3, 4, np.random.RandomState(11)),
    "reg_4_8": lambda codes: codes.rand_reg_ldpc(1200, 4, 8, np.random.RandomState(12)),
    "reg_3_5": lambda codes: codes.rand_reg_ldpc(1200, 3, 5, np.random.RandomState(13)),
    "rho_x4": lambda codes: codes.rand_irregular_ldpc(1200, codes.LAMBDA_HALF_RATE[4], 5, np.random.RandomState(14)),
    "rho_x6": lambda codes: codes.rand_irregular_ldpc(1200, codes.LAMBDA_HALF_RATE[6], 7, np.random.RandomState(15)),
}
SNR = {"reg_3_4": -1.0, "reg_4_8": 2.4, "reg_3_5": 0.6, "rho_x4": 1.6, "rho_x6": 1.4}  # around each code's waterfall: a mix of exits
_cache = {}


def _code(name):
    from ldpc_decoders_amd import codes

    if name not in _cache:
        code = CODES[name](codes)
        _cache[name] = (O.Edges(code.m, code.n, code.edge_chk, code.edge_var), code)
    return _cache[name]


def _priors(g, snr, B, seed):
    rng = np.random.RandomState(seed)
    return O.biawgn_priors(-1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(snr)), (B, g.n)), snr)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("name", sorted(CODES))
def test_min_sum_bit_exact_on_the_fused_kernels(name, prec):
    from ldpc_decoders_amd import bpa

    g, code = _code(name)
    dt = np.float64 if prec == "f64" else np.float32
    dec = bpa.MSA(code, max_iter=50, precision=prec, backend="fused")
    fi = dec.handle.fused_info()
    assert fi["waves_per_frame"] == 2
    for B, seed in ((5, 1), (257, 2)):
        pri = _priors(g, SNR[name], B, seed).astype(dt)
        xhat, iters = dec.decode_batch(None, pri)
        assert dec.handle.last_stats()[0] == "fused"
        xo, io = C.bp_decode(g, "MSA", None, pri, 50, dtype=dt)
        assert (xhat == xo).all() and (iters == io).all()
    assert len(np.unique(io)) > 3  # early exits and capped frames both present
    print("%s %s: %d checks of degree %s, variable degrees up to %d, LDS %d B/frame, %.0f conflict cycles/sweep (identity %.0f)" % (
        name, prec, g.m, sorted(set(code.row_degrees().tolist())), code.col_degrees().max(), fi["lds_bytes_per_frame"], fi["conflict_cycles_planned"],
        fi["conflict_cycles_identity"]))


@pytest.mark.parametrize("name", sorted(CODES))
def test_erasure_decoder_exact_on_the_fused_kernels(name):
    import torch
    from ldpc_decoders_amd._device import DecoderHandle

    g, code = _code(name)
    h = DecoderHandle(code, "BEC", "f32", "fused")
    eps = {"reg_3_4": 0.62, "reg_4_8": 0.36, "reg_3_5": 0.48, "rho_x4": 0.42, "rho_x6": 0.44}[name]
    _, y = h.channel_device("bec", eps, 0, 5, 1, 0, 700)
    for mi in (50, 3):
        xh, it = h.decode_device(None, y, mi)
        xo, io = C.bec_decode(g, y.cpu().numpy(), mi)
        assert (xh.cpu().numpy() == xo).all() and (it.cpu().numpy() == io).all()
    cnt = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
    h.simulate("bec", eps, 0, 5, 1, 0, 700, 50, cnt, hist_bins=51)  # channel + decode + count in one kernel
    xo, io = C.bec_decode(g, y.cpu().numpy(), 50)
    err = (xo != 0).sum(axis=1)
    assert cnt[:4].cpu().tolist() == [700, int((err > 0).sum()), int(err.sum()), int(io.sum())]
    assert 0 < (err > 0).sum() < 700


@pytest.mark.parametrize("name", sorted(CODES) + ["1200_3_6_rand_ldpc_1", "1200_rho_x5_rand_ldpc_5", "margulis"])
def test_fp64_sum_product_on_the_lds_equals_the_streaming_kernel(name):
    # the reference's sum-product arithmetic (src/bpa.py:66-75 verbatim) through the same device function on both backends, the row
    # sum of logs in the reference's edge order on both: decisions, iteration counts AND marginals bit-identical
    import torch
    from helpers import golden_edges
    from ldpc_decoders_amd import bpa
    from ldpc_decoders_amd.codes import Code

    if name in CODES:
        g, code = _code(name)
        snr = SNR[name]
    else:
        g = golden_edges(name)
        code = Code.from_edges(g.m, g.n, g.chk, g.var)
        snr = 1.8
    pri = torch.from_numpy(_priors(g, snr, 150, 9)).cuda()
    outs = []
    for backend in ("stream", "fused"):
        dec = bpa.SPA(code, max_iter=40, precision="f64", backend=backend)
        x, it, mg = dec.handle.decode_soft_device(pri, None, 40)
        assert dec.handle.last_stats()[0] == backend
        outs.append((x.cpu().numpy(), it.cpu().numpy(), mg.cpu().numpy()))
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()
    same = (outs[0][2] == outs[1][2]) | (np.isnan(outs[0][2]) & np.isnan(outs[1][2]))
    assert same.all()
    assert len(np.unique(outs[0][1])) > 3


@pytest.mark.parametrize("name", sorted(CODES))
def test_fp32_sum_product_on_the_fused_kernels(name):
    # fp32 sum-product (phi-domain rule): decisions against the fp64 statement of the same rule (numpy oracle) -- >= 95 % of frames
    # identical, as for the degree-6 shapes (tests/test_gpu_parity.py::test_spa_fp32_decisions) -- and against the streaming kernel
    from ldpc_decoders_amd import bpa

    g, code = _code(name)
    pri = _priors(g, SNR[name], 48, 17)
    dec = bpa.SPA(code, max_iter=30, precision="f32", backend="fused")
    xhat, iters = dec.decode_batch(None, pri.astype(np.float32))
    xo, io = O.bp_decode(g, "SPA_PHI", -pri, pri, 30)  # (y only matters for the iteration-0 test: real valued, never passes)
    same = (xhat == xo).all(axis=1)
    xs, its = bpa.SPA(code, max_iter=30, precision="f32", backend="stream").decode_batch(None, pri.astype(np.float32))
    same_s = (xhat == xs).all(axis=1)
    print("%s: fused fp32 sum-product == fp64 phi oracle on %.1f %% of frames, == streaming kernel on %.1f %%" % (name, 100 * same.mean(), 100 * same_s.mean()))
    assert same.mean() >= 0.95 and same_s.mean() >= 0.95
    assert (np.abs(iters - io) <= 1)[io < 30].mean() >= 0.9


@pytest.mark.parametrize("alg,channel,param", [("MSA", "biawgn", 2.0), ("MSA", "bsc", 0.05), ("SPA", "biawgn", 1.6), ("SPA", "bsc", 0.07)])
@pytest.mark.parametrize("name", ["1200_3_6_rand_ldpc_1", "1200_rho_x5_rand_ldpc_5", "reg_4_8", "rho_x6"])
def test_fp64_in_kernel_channel_equals_the_channel_kernel(name, alg, channel, param):
    # fp64 simulate = ONE kernel (Philox noise -> LLR -> decode -> count, priors never in HBM).  Its counters must equal the
    # composition channel kernel -> HBM -> ldpc_decode -> ldpc_count_errors frame for frame (same inline noise functions).
    import torch
    from helpers import golden_edges
    from ldpc_decoders_amd import _lib
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    if name in CODES:
        g, code = _code(name)
    else:
        g = golden_edges(name)
        code = Code.from_edges(g.m, g.n, g.chk, g.var)
    h = DecoderHandle(code, alg, "f64", "fused")
    B, frame0 = 1500, 123456
    for cw in (0, 1):
        cnt = torch.zeros(4 + 41, dtype=torch.int64, device="cuda")
        h.simulate(channel, param, cw, 99, 3, frame0, B, 40, cnt, hist_bins=41)
        pri, y = h.channel_device(channel, param, cw, 99, 3, frame0, B)
        xh, it = h.decode_device(pri, y, 40)
        ref = torch.zeros(4 + 41, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().ldpc_count_errors(xh.data_ptr(), None, cw, it.data_ptr(), B, g.n, 41, ref.data_ptr(), st))
        assert cnt.cpu().tolist() == ref.cpu().tolist()
        assert cnt[0].item() == B and cnt[3].item() > B


def test_streaming_backend_high_degrees():
    # the streaming kernels take node degrees up to 64 (instantiated maxima 4, 6, 8, 16, 32, 64): a (5,20)-regular code, a (3,33)-regular one
    # (odd degree beyond 32) and a hand-made matrix with check degrees spread from 2 to 40 and variable degrees from 1 to 24, all three
    # decoders and both arithmetics against the C oracle
    import bp_oracle as O
    import c_oracle as C
    from ldpc_decoders_amd import bpa, bec, codes
    from ldpc_decoders_amd.codes import Code

    rng = np.random.RandomState(12)
    cases = [codes.rand_reg_ldpc(2000, 5, 20, rng), codes.rand_reg_ldpc(1980, 3, 33, rng)]
    # spread degrees: check i takes (2 + i % 39) distinct variables drawn with a skew towards low indices
    n, m = 900, 300
    rows = []
    for i in range(m):
        d = 2 + i % 39
        p = 1.0 / (1.0 + np.arange(n) / 40.0)
        rows.append(sorted(rng.choice(n, size=d, replace=False, p=p / p.sum()).tolist()))
    chk = np.concatenate([[i] * len(r) for i, r in enumerate(rows)]).astype(np.int32)
    var = np.concatenate(rows).astype(np.int32)
    spread = Code.from_edges(m, n, chk, var)
    assert spread.row_degrees().max() == 40 and spread.col_degrees().max() >= 20 and spread.col_degrees().min() <= 1
    cases.append(spread)
    for code in cases:
        class G:
            m, n, chk, var = code.m, code.n, code.edge_chk, code.edge_var

        B = 130
        y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(4.0)), (B, code.n))
        pri = O.biawgn_priors(y, 4.0)
        for prec, dt in (("f64", np.float64), ("f32", np.float32)):
            dec = bpa.MSA(code, max_iter=12, precision=prec, backend="auto")
            xhat, iters = dec.decode_batch(None, pri.astype(dt))
            assert dec.handle.last_stats()[0] == "stream"
            xo, io = C.bp_decode(G, "MSA", None, pri.astype(dt), 12, dtype=dt)
            assert (xhat == xo).all() and (iters == io).all()
        ys = (rng.random_sample((B, code.n)) < 0.25).astype(np.uint8) * 2  # erasures of the all-zero word
        xb, ib = bec.SPA(0.25, code, max_iter=12).decode_batch(ys)
        xo, io = C.bec_decode(G, ys, 12)
        assert (xb == xo).all() and (ib == io).all()


def test_empty_check_rows_including_trailing_ones():
    # the branch-free ("dense") line fetch of the check pass computed an edge index for EVERY row; an empty row at the end of
    # H has row_ptr == E, one past the edge list.  (3,4)-, rho = x^5-like and (3,6)-shaped toy matrices with empty rows in the middle and
    # at the end, all three decoders, both arithmetics, against the C oracle; a frame is a codeword exactly when the non-empty checks hold.
    import bp_oracle as O
    import c_oracle as C
    from ldpc_decoders_amd import bec, bpa, codes
    from ldpc_decoders_amd.codes import Code

    rng = np.random.RandomState(21)
    for base, extra in ((codes.rand_reg_ldpc(240, 3, 4, rng), 3), (codes.rand_reg_ldpc(600, 3, 6, rng), 2), (codes.get_code("12_3_4_ldpc"), 5)):
        # two empty rows spliced into the middle, `extra` at the end
        mid = base.m // 2
        chk = np.where(base.edge_chk >= mid, base.edge_chk + 2, base.edge_chk)
        code = Code.from_edges(base.m + 2 + extra, base.n, chk, base.edge_var)
        assert code.row_degrees()[-extra:].sum() == 0 and code.row_degrees()[mid:mid + 2].sum() == 0

        class G:
            m, n, chk, var = code.m, code.n, code.edge_chk, code.edge_var

        B = 200
        y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(3.0)), (B, code.n))
        pri = O.biawgn_priors(y, 3.0)
        for prec, dt in (("f64", np.float64), ("f32", np.float32)):
            dec = bpa.MSA(code, max_iter=15, precision=prec, backend="stream")
            xhat, iters = dec.decode_batch(None, pri.astype(dt))
            xo, io = C.bp_decode(G, "MSA", None, pri.astype(dt), 15, dtype=dt)
            assert (xhat == xo).all() and (iters == io).all()
            assert len(np.unique(iters)) > 1 and code.syndrome(xhat[iters < 15]).sum() == 0
        dec = bpa.SPA(code, max_iter=15, precision="f64", backend="stream")
        xhat, iters = dec.decode_batch(None, pri)
        xo, io = C.bp_decode(G, "SPA", None, pri, 15)
        assert (xhat == xo).mean() > 0.999 and (np.abs(iters - io) <= 1).all()
        ye = (rng.random_sample((B, code.n)) < 0.3).astype(np.uint8) * 2
        de = bec.SPA(0.3, code, max_iter=15, backend="stream")
        xe, ie = de.decode_batch(ye)
        xo, io = C.bec_decode(G, ye, 15)
        assert (xe == xo).all() and (ie == io).all()
