"""Parity at the sizes of BASELINE.json configs 4 and 5 (the reference itself cannot run these: its H is dense, SURVEY.md 8(c)
'limits of the oracle').  Checker = the C oracle, pinned to the reference at n <= 2640."""
import os

import numpy as np
import pytest

import bp_oracle as O
import c_oracle as C

pytestmark = pytest.mark.gpu


class _G:  # minimal graph view for the C oracle
    def __init__(self, code):
        self.m, self.n, self.chk, self.var = code.m, code.n, code.edge_chk, code.edge_var


def _noise(rng, B, n, snr):
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(snr)), (B, n))
    return O.biawgn_priors(y, snr)


def test_config4_irregular_n10000_msa():
    # rate-1/2 irregular ensemble of src/ldpc.py (lambda from its LP design, rho = x^5), n = 10 000, min-sum: the streaming
    # backend (fp32 / fp64) and the 16-wave LDS-resident kernel (fp32, one frame per CU) against the C oracle
    from ldpc_decoders_amd import bpa, codes

    code = codes.rand_irregular_ldpc(10000, codes.LAMBDA_RHO_X5_HALF_RATE, 6, np.random.RandomState(4))
    assert code.n == 10000 and code.col_degrees().max() == 8 and set(np.unique(code.row_degrees())) <= {2, 4, 6}
    pri = _noise(np.random.RandomState(1), 192, code.n, 1.6)
    for prec, dt, backend, used in (("f32", np.float32, "stream", "stream"), ("f64", np.float64, "auto", "stream"),
                                    ("f32", np.float32, "auto", "fused")):
        dec = bpa.MSA(code, max_iter=50, precision=prec, backend=backend)
        xhat, iters = dec.decode_batch(None, pri.astype(dt))
        xo, io = C.bp_decode(_G(code), "MSA", None, pri.astype(dt), 50, dtype=dt)
        assert dec.handle.last_stats()[0] == used
        assert (xhat == xo).all() and (iters == io).all()
        assert 1 < iters.mean() < 50 and (iters < 50).any() and (iters == 50).any()  # converging and failing frames present
    info = dec.handle.fused_info()
    assert info["waves_per_frame"] == 16 and info["lds_bytes_per_frame"] == 160 * 1024


def test_config4_shipped_plan_builds_tables_with_few_conflicts(monkeypatch, tmp_path):
    # bench.py's config-4 code with its shipped plan: the bank-conflict cycles per sweep of the gather tables AS BUILT -- every lane's final
    # address, i.e. the padding reads of short check rows, of variables below their round's width and of padded check lanes included --
    # stay close to what the planner reports for the real edges alone: there is a "certain" slot / a zero word on every bank and each
    # half-wave reads the one on the bank it leaves free.  (Rounds 3-5 had ONE word of each kind: 770 cycles on top of the planner's 65,
    # which the PMC pass of the 16-wave kernel measured as SQ_LDS_BANK_CONFLICT 787 per frame-sweep.)
    import ctypes

    import bench
    from ldpc_decoders_amd import _lib, bpa

    monkeypatch.setenv("LDPC_FUSED_PLAN_SAVE", "none")
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "empty"))
    code = bench.load_code("gen:irg:10000")
    info = bpa.MSA(code, max_iter=50, precision="f32", backend="fused").handle.fused_info()
    assert info["waves_per_frame"] == 16 and info["lds_gather_cycles_min"] == 2 * (80 * 6 + 16 * 34)  # the shape with pair rounds
    plan = (ctypes.c_double * 4)()  # host-only planner entry: {-waves (stored plan), gather cycles, trivial placement, plan: real edges}
    _lib.check(_lib.load().ldpc_plan_layout(code.m, code.n, code.E, np.ascontiguousarray(code.edge_chk, dtype=np.int32).ctypes.data,
                                            np.ascontiguousarray(code.edge_var, dtype=np.int32).ctypes.data, 0, 0, -1, None, plan))
    assert plan[0] == -16 and plan[3] < 0.08 * plan[2]
    assert plan[3] <= info["conflict_cycles_planned"] <= plan[3] + 48, (info, list(plan))


def test_config4_fused_layout_independent(monkeypatch):
    # the trivial placement and a short annealing run give the same bits as the stored plan (placement only moves data)
    from ldpc_decoders_amd import bpa, codes

    code = codes.rand_irregular_ldpc(10000, codes.LAMBDA_RHO_X5_HALF_RATE, 6, np.random.RandomState(4))
    pri = _noise(np.random.RandomState(5), 70, code.n, 1.6).astype(np.float32)
    xo, io = C.bp_decode(_G(code), "MSA", None, pri, 30, dtype=np.float32)
    seen = []
    for layout, moves in (("identity", None), ("replan", "300000")):
        monkeypatch.setenv("LDPC_FUSED_LAYOUT", layout)
        if moves:
            monkeypatch.setenv("LDPC_FUSED_PLAN_MOVES", moves)
        dec = bpa.MSA(code, max_iter=30, precision="f32", backend="fused")
        xhat, iters = dec.decode_batch(None, pri)
        assert (xhat == xo).all() and (iters == io).all()
        seen.append(dec.handle.fused_info()["conflict_cycles_planned"])
    assert seen[0] > seen[1] > 0


def test_config4_fused_erasure_sum_product_and_simulate():
    # the other decoders at n = 10 000: erasure decoder (exact vs the C oracle, stopping sets included; a slab of 32 frames of this code
    # is 245 KB of bit planes, more than a CU's LDS, so the bit-sliced erasure decoder runs it on the streaming kernels), sum-product on
    # the 16-wave shape (same arithmetic as the streaming kernels), and the fused simulate kernels (noise + decode + count) vs the streaming path
    import torch
    from ldpc_decoders_amd import bec, bpa, codes
    from ldpc_decoders_amd._device import DecoderHandle

    code = codes.rand_irregular_ldpc(10000, codes.LAMBDA_RHO_X5_HALF_RATE, 6, np.random.RandomState(4))
    rng = np.random.RandomState(3)
    ye = (rng.random_sample((130, code.n)) < 0.44).astype(np.int64) * 2
    ye[:3, :] = np.where(ye[:3, :] == 0, 1, 2)  # a few frames of the all-one word
    ye[3] = 0
    for mi in (50, 3):
        dec = bec.SPA(0.44, code, max_iter=mi)
        xe, ie = dec.decode_batch(ye)
        xo, io = C.bec_decode(_G(code), ye, mi)
        assert dec.handle.last_stats()[0] == "stream" and (xe == xo).all() and (ie == io).all()
    assert (ie[:50] < 50).any()
    pri = _noise(rng, 130, code.n, 1.7).astype(np.float32)
    a, ia = bpa.SPA(code, max_iter=50, precision="f32", backend="fused").decode_batch(None, pri)
    b, ib = bpa.SPA(code, max_iter=50, precision="f32", backend="stream").decode_batch(None, pri)
    assert ((a == b).all(axis=1)).mean() >= 0.97 and (ia == ib).mean() >= 0.97
    for alg, ch, prm in (("MSA", "biawgn", 1.6), ("BEC", "bec", 0.44), ("MSA", "bsc", 0.06)):
        res = []
        for be in ("fused", "stream"):
            h = DecoderHandle(code, alg, "f32", "auto" if (alg, be) == ("BEC", "fused") else be)
            cnt = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
            h.simulate(ch, prm, 0, 7, 1, 1000, 300, 50, cnt, hist_bins=51)
            assert h.last_stats()[0] == ("stream" if alg == "BEC" else be)
            res.append(cnt.cpu().numpy())
        assert (res[0] == res[1]).all() and res[0][0] == 300 and 0 < res[0][1] < 300


def test_config5_regular_n64800_msa_early_termination():
    from ldpc_decoders_amd import bpa, codes

    code = codes.rand_reg_ldpc(64800, 3, 6, np.random.RandomState(8))
    assert (code.m, code.E) == (32400, 194400)
    pri = _noise(np.random.RandomState(2), 96, code.n, 1.9).astype(np.float32)
    dec = bpa.MSA(code, max_iter=50, precision="f32")
    xhat, iters = dec.decode_batch(None, pri)
    xo, io = C.bp_decode(_G(code), "MSA", None, pri, 50, dtype=np.float32)
    assert (xhat == xo).all() and (iters == io).all()
    done = iters < 50
    assert done.any() and code.syndrome(xhat[done]).sum() == 0


def _published_point(channel, code_name, decoder, max_iter, codeword, param):
    """One point of a result file the reference publishes (data/output/*.json, kept as tests/golden/published_curves.json)."""
    import json

    from helpers import GOLDEN

    with open(os.path.join(GOLDEN, "published_curves.json")) as fp:
        for e in json.load(fp):
            if (e["channel"], e["code"], e["decoder"], str(e.get("max_iter")), str(e.get("codeword"))) == (channel, code_name, decoder, str(max_iter), str(codeword)):
                return e["points"][param]
    raise KeyError((channel, code_name, decoder, max_iter, codeword, param))


def test_config3_spa_bsc_full_batch(tmp_path):
    # config 3: n = 1200 sum-product over the BSC, batch 65 536 on one GPU (device channel kernel, fp32 LDS kernel).
    # (1) size-independent properties on the whole batch; (2) ALL 65 536 frames re-decoded by the fp64 phi-domain oracle: identical decisions
    # (measured: all but a few non-converging frames, which are named below); (3) the reference's own arithmetic (fp64, formula
    # verbatim) at the reference's published operating point -- bsc-1200_3_6_rand_ldpc_1-SPA-10-0.json, p = 0.06, 581 frames upstream --
    # word-error rate within 4 sigma of the published value, bit-error rate within its spread.
    import subprocess
    import sys

    import torch
    from helpers import golden_edges
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges("1200_3_6_rand_ldpc_1")
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    B = 65536
    h = DecoderHandle(code, "SPA", "f32")
    pri, y = h.channel_device("bsc", 0.06, 0, 11, 0, 0, B)
    xhat, iters = h.decode_device(pri, y, 50)
    xh, it = xhat.cpu().numpy(), iters.cpu().numpy()
    done = it < 50
    assert done.mean() > 0.9 and code.syndrome(xh[done]).sum() == 0     # every frame that left early carries a codeword
    idx = np.arange(B)                                                   # EVERY frame of the batch (round 3 re-decoded 4 096 of them)
    yh, ph = y[idx].cpu().numpy().astype(float), pri[idx].double().cpu().numpy()
    # the oracle runs in a fresh process (tests/phi_redecode.py forks its worker pool there, not under this process's GPU runtime)
    src, dst = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(src, y=yh, pri=ph)
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "phi_redecode.py"), src, dst, "1200_3_6_rand_ldpc_1", "50",
                    str(min(os.cpu_count() or 16, 128))], check=True, timeout=900)
    res = np.load(dst)
    xo, io = res["x"], res["it"]
    differ = np.flatnonzero(~(xh[idx] == xo).all(axis=1))
    print("config 3 sum-product / BSC: %d of %d re-decoded frames differ from the fp64 phi oracle: %s" % (len(differ), len(idx), idx[differ].tolist()))
    # fp32 arithmetic against the fp64 statement of the same rule: a frame may end in a different word only at the edge of convergence --
    # frames the fp64 oracle itself needs at least 15 sweeps for (the batch's mean is about 7; measured: every differing frame needs 19
    # or more, half of them never converge).  There the iteration is chaotic and the last word depends on the last bit of every message:
    # WHICH of those frames differ changes with any re-association of the arithmetic (round 4's prefix / suffix form and round 5's pair
    # tree in the base-2 domain each differ on 26 of the 65 536, 23 of them the same frames).  Every other frame must be identical.
    slow = io >= 15
    print("config 3 sum-product / BSC: oracle sweeps of the differing frames %s; %d frames of the batch need >= 15 sweeps (mean %.2f)" % (
        sorted(io[differ].tolist()), int(slow.sum()), float(io.mean())))
    assert slow[differ].all(), "a frame that converges quickly differs from the fp64 oracle"
    # ... and the GPU's own run of those frames is just as slow: a regression that broke slow-converging but decodable frames would show
    # as differing frames the GPU leaves early (or as many more of them than the 26 +- a few that re-association moves around)
    assert (it[idx][differ] >= 15).all(), "the fp32 kernel leaves a differing frame early: %s" % sorted(it[idx][differ].tolist())
    assert len(differ) <= 32, "%d frames differ (rounds 4 and 5: 26)" % len(differ)
    assert (np.abs(it[idx] - io) <= 1)[io < 50].mean() >= 0.999
    # the published curve, in the reference's arithmetic
    ref = _published_point("bsc", "1200_3_6_rand_ldpc_1", "SPA", 10, 0, "0.06")
    h64 = DecoderHandle(code, "SPA", "f64")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    h64.simulate("bsc", 0.06, 0, 11, 3, 0, B, 10, cnt)
    tot, wec, bec = (int(v) for v in cnt.cpu().numpy()[:3])
    wer, ber = wec / tot, bec / (tot * code.n)
    sigma = np.sqrt(ref["wer"] * (1 - ref["wer"]) * (1 / ref["tot"] + 1 / tot))
    print("config 3 sum-product / BSC p=0.06 max_iter=10 fp64: WER %.4f (published %.4f +- %.4f), BER %.3e (published %.3e)" % (wer, ref["wer"], sigma, ber, ref["ber"]))
    assert abs(wer - ref["wer"]) <= 4 * sigma
    # bit errors per failed word: 100 failed words upstream, spread of the count per word about its mean -> 4 sigma ~ 40 %
    assert 0.6 <= (ber / wer) / (ref["ber"] / ref["wer"]) <= 1.6


def test_config3_erasure_full_batch():
    # config 3, erasure decoder over the BEC: EVERY frame of the 65 536-frame batch against the C oracle (decisions and iteration
    # counts), and the reference's published points at its own iteration cap (bec-1200_3_6_rand_ldpc_1-SPA-10-0.json) within 4 sigma.
    import torch
    from helpers import golden_edges
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges("1200_3_6_rand_ldpc_1")
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    B = 65536
    hb = DecoderHandle(code, "BEC", "f32")
    _, ye = hb.channel_device("bec", 0.40, 0, 12, 0, 0, B)
    xe, ie = hb.decode_device(None, ye, 50)
    xe_h, ye_h = xe.cpu().numpy(), ye.cpu().numpy()
    assert ((xe_h == ye_h) | (ye_h == 2)).all()  # known symbols are never changed
    assert ((xe_h == 2).sum(axis=1) <= (ye_h == 2).sum(axis=1)).all() and (xe_h[xe_h != 2] == 0).all()
    xo, io = C.bec_decode(g, ye_h, 50)
    assert (xe_h == xo).all() and (ie.cpu().numpy() == io).all()
    for eps in ("0.4", "0.375"):
        ref = _published_point("bec", "1200_3_6_rand_ldpc_1", "SPA", 10, 0, eps)
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        hb.simulate("bec", float(eps), 0, 12, 5, 0, B, 10, cnt)
        tot, wec, bec = (int(v) for v in cnt.cpu().numpy()[:3])
        wer, ber = wec / tot, bec / (tot * code.n)
        sigma = np.sqrt(ref["wer"] * (1 - ref["wer"]) * (1 / ref["tot"] + 1 / tot))
        print("config 3 erasure decoder eps=%s max_iter=10: WER %.4f (published %.4f +- %.4f), BER %.3e (published %.3e)" % (eps, wer, ref["wer"], sigma, ber, ref["ber"]))
        assert abs(wer - ref["wer"]) <= 4 * sigma
        assert 0.6 <= (ber / wer) / (ref["ber"] / ref["wer"]) <= 1.6


def test_config4_fused_deterministic():
    import torch
    from ldpc_decoders_amd import codes
    from ldpc_decoders_amd._device import DecoderHandle

    code = codes.rand_irregular_ldpc(10000, codes.LAMBDA_RHO_X5_HALF_RATE, 6, np.random.RandomState(4))
    h = DecoderHandle(code, "MSA", "f32", "fused")
    pri, _ = h.channel_device("biawgn", 1.7, 0, 5, 0, 0, 1500)
    x0, i0 = h.decode_device(pri, None, 50)
    x0, i0 = x0.clone(), i0.clone()
    assert len(torch.unique(i0)) > 3
    for _ in range(4):
        x1, i1 = h.decode_device(pri, None, 50)
        assert torch.equal(x1, x0) and torch.equal(i1, i0)


def test_full_size_configs_4_and_5_properties():
    # BASELINE configs 4 and 5 at their FULL per-GPU sizes (2^20 / 8 = 131 072 frames of the rate-1/2 irregular n = 10 000 ensemble;
    # 2^18 / 8 = 32 768 frames of the (3,6) n = 64 800 code), device noise.  The oracle cannot run these sizes, so the checks are
    # size-independent properties: the counters of the whole batch equal the sum over two shards (frames are keyed by their global
    # index: what makes the 8-GPU split exact), every frame that left early carries a codeword (zero syndrome), the all-zero word
    # is recovered at this SNR, iteration counts are spread (per-frame early termination at work, with the streaming backend
    # re-forming its tiles), and a sample of frames re-decoded by the C oracle is bit-identical.
    import torch
    from ldpc_decoders_amd import codes
    from ldpc_decoders_amd._device import DecoderHandle

    for name, code, B, snr, backend in (("config 5", codes.rand_reg_ldpc(64800, 3, 6, np.random.RandomState(20261002)), 32768, 2.0, "stream"),
                                         ("config 4", codes.rand_irregular_ldpc(10000, codes.LAMBDA_RHO_X5_HALF_RATE, 6, np.random.RandomState(20261002)), 131072, 2.1, "fused")):
        h = DecoderHandle(code, "MSA", "f32", backend)

        def counters(f0, nb):
            c = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
            h.simulate("biawgn", snr, 0, 99, 0, f0, nb, 50, c, hist_bins=51)
            torch.cuda.synchronize()
            return c.cpu().numpy()

        whole = counters(0, B)
        assert h.last_stats()[0] == backend
        half = counters(0, B // 2 + 7) + counters(B // 2 + 7, B - B // 2 - 7)
        assert (whole == half).all(), name
        hist = whole[4:]
        assert whole[0] == B and hist.sum() == B and (hist > 0).sum() >= 8, (name, hist)  # frames leave after many different sweep counts
        assert whole[1] <= 0.4 * B, (name, whole[:4])  # most frames decode at this SNR
        # a slice of the same frames through the decode entry: codewords where the frame left early, oracle-identical on a sample
        nb = 1024 if backend == "stream" else 4096
        pri, _ = h.channel_device("biawgn", snr, 0, 99, 0, 0, nb)
        xhat, iters = h.decode_device(pri, None, 50)
        xh, it = xhat.cpu().numpy(), iters.cpu().numpy()
        early = it < 50
        assert early.mean() > 0.8 and code.syndrome(xh[early][:256]).sum() == 0
        assert (np.bincount(it, minlength=51) == counters(0, nb)[4:]).all()  # the decode entry and the Monte-Carlo path agree frame for frame
        if backend == "stream":
            assert h.last_repacks() >= 1

        class G:
            m, n, chk, var = code.m, code.n, code.edge_chk, code.edge_var

        idx = np.r_[0:8, nb - 8:nb]
        xo, io = C.bp_decode(G, "MSA", None, pri[idx].cpu().numpy(), 50, dtype=np.float32)
        assert (xh[idx] == xo).all() and (it[idx] == io).all(), name
        del h, pri, xhat, iters
        torch.cuda.empty_cache()
