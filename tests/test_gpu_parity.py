"""GPU parity tests: the HIP path (through the C ABI) against the reference's golden vectors and the CPU oracle.

Bars: min-sum (fp64, and fp32 on exactly representable priors) and the erasure decoder are BIT-EXACT in hard decisions
and iteration counts; sum-product is held to a stated tolerance (libm / fp32 rounding differs from numpy's).
"""
import os

import numpy as np
import pytest

import bp_oracle as O
import c_oracle as C
from helpers import case_id, decode_cases, expected_xhat, golden_edges, kat_cases, load_case

pytestmark = pytest.mark.gpu

BACKENDS = ["stream", "auto"]


def _code(name):
    from ldpc_decoders_amd.codes import Code

    g = golden_edges(name)
    return g, Code.from_edges(g.m, g.n, g.chk, g.var)


def _priors(c):
    if c["channel"] == "biawgn":
        return O.biawgn_priors(c["y"].astype(np.float64), c["param"])
    return O.bsc_priors(c["y"].astype(np.int64), c["param"])


@pytest.mark.parametrize("kat", kat_cases(), ids=lambda k: "%s-%s-%s" % (k["channel"], k["code"], k["decoder"]))
def test_known_answer_through_registry(kat):
    # the reference's own smoke tests (src/biawgn.py:81-92, src/bsc.py:78-89, src/bec.py:128-139), same call shape as
    # utils.TestCase.sample (src/utils.py:84): decoder(param, code, **kwargs).decode(y_)
    from ldpc_decoders_amd.models import models

    _, code = _code(kat["code"])
    dec = getattr(models[kat["channel"]], kat["decoder"])(kat["param"], code, max_iter=kat["max_iter"], mu=3., eps=1e-5, allow_pseudo=1)
    est = dec.decode(np.array(kat["received"]))
    assert (np.asarray(est) == np.array(kat["sent"])).all()
    assert (np.asarray(est, dtype=float) == np.array(kat["reference_estimate"])).all()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("path", decode_cases("*_MSA_*"), ids=case_id)
def test_msa_fp64_bit_exact_vs_reference(path, backend):
    from ldpc_decoders_amd import bec, bpa

    c = load_case(path)
    g, code = _code(c["code"])
    want = expected_xhat(c)
    if c["channel"] == "bec":
        dec = bec.MSA(c["param"], code, max_iter=c["max_iter"], backend=backend)
        xhat, _ = dec.decode_batch(c["y"])
        assert (xhat == want).all()
        return
    dec = bpa.MSA(code, max_iter=c["max_iter"], precision="f64", backend=backend)
    y0 = None if c["channel"] == "biawgn" else c["y"]
    xhat, iters = dec.decode_batch(y0, _priors(c))
    keep = np.setdiff1d(np.arange(c["nframes"]), c["raw_rows"])
    assert (xhat[keep] == want[keep]).all()
    assert (iters[keep] == c["iters"][keep]).all()


@pytest.mark.parametrize("path", decode_cases("bec_*"), ids=case_id)
def test_bec_exact_vs_reference(path):
    from ldpc_decoders_amd import bec

    c = load_case(path)
    g, code = _code(c["code"])
    dec = bec.SPA(c["param"], code, max_iter=c["max_iter"])
    xhat, iters = dec.decode_batch(c["y"])
    assert (xhat == expected_xhat(c)).all()
    _, it_o = C.bec_decode(g, c["y"], c["max_iter"])
    assert (iters == it_o).all()
    one = dec.decode(c["y"][0])  # single-frame registry call
    assert (one == expected_xhat(c)[0]).all()


# Frames that MAY differ from the reference in fp64 sum-product, by (case id, backend): device libm differs from numpy's by ulps, which
# can flip a chaotic non-converging frame.  MEASURED (every golden case x both backends, profiles/r03H_parity_measured.txt): no frame
# differs anywhere, so the list is empty and the test demands identity; a kernel change that flips a frame must name it here.
SPA_F64_ALLOWED_FLIPS = {}


@pytest.mark.parametrize("backend", ["stream", "auto"])
@pytest.mark.parametrize("path", [p for p in decode_cases("*_SPA_*") if "bec_" not in os.path.basename(p)], ids=case_id)
def test_spa_fp64_vs_reference(path, backend):
    # fp64 SPA follows the reference formula; held to what is measured: every frame's decisions identical to the reference (allow-list
    # above), iteration counts of the frames that converge upstream identical too.  Both backends.  The measured agreement of every
    # case is printed (pytest -s; profiles/r03H_parity_measured.txt keeps the lines of the round).
    from ldpc_decoders_amd import bpa

    c = load_case(path)
    g, code = _code(c["code"])
    dec = bpa.SPA(code, max_iter=c["max_iter"], precision="f64", backend=backend)
    y0 = None if c["channel"] == "biawgn" else c["y"]
    xhat, iters = dec.decode_batch(y0, _priors(c))
    want = expected_xhat(c)
    keep = np.setdiff1d(np.arange(c["nframes"]), c["raw_rows"])
    same = (xhat[keep] == want[keep]).all(axis=1)
    conv = c["iters"][keep] < c["max_iter"]
    it_same = iters[keep][conv] == c["iters"][keep][conv]
    print("fp64 sum-product vs reference, %s backend=%s(%s): frames identical %d/%d (%.1f %%), of the %d converging upstream %d (%.1f %%), "
          "iteration counts identical on %.1f %% of those" % (case_id(path), backend, dec.handle.last_stats()[0], same.sum(), len(same), 100 * same.mean(),
                                                               conv.sum(), same[conv].sum(), 100 * same[conv].mean() if conv.any() else 100.0,
                                                               100 * it_same.mean() if conv.any() else 100.0))
    allowed = SPA_F64_ALLOWED_FLIPS.get((case_id(path), backend), set())
    differing = set(int(f) for f in keep[~same])
    assert differing <= allowed, "frames %s differ from the reference (allowed: %s)" % (sorted(differing - allowed), sorted(allowed))
    # iteration counts of every converging frame that is not on the allow-list (one allow-listed flip must not switch the
    # check off for all the others)
    checked = ~np.isin(keep[conv], sorted(allowed))
    assert it_same[checked].all(), "iteration counts of converging frames differ: %s" % keep[conv][checked][~it_same[checked]]


SPA_TRACE_CASES = [p for p in decode_cases("*_SPA_*") if "bec_" not in p]
MSA_TRACE_CASES = [p for p in decode_cases("*_MSA_*") if "bec_" not in p]
FUSED_CODES = ("1200_3_6", "1200_rho_x5", "512_3_6", "margulis")  # codes with an LDS-resident kernel shape

# (algorithm, arithmetic, backend, tolerance): |dev - ref| <= rtol * (1 + |ref|) on the marginal LLRs -- the north_star bar is on
# SOFT values, so every kernel that carries throughput is held to it, not only the streaming one.  Min-sum in fp64 must be
# bit-identical (rtol 0: add/sub/compare only); fp32 min-sum differs by fp32 rounding of the sums; sum-product by libm / the
# fp32 phi-domain rule.
SOFT_MODES = [("SPA", "f64", "stream", 1e-9), ("SPA", "f64", "fused", 1e-9), ("SPA", "f32", "stream", 2e-3), ("SPA", "f32", "fused", 2e-3),
              ("MSA", "f64", "stream", 0.0), ("MSA", "f64", "fused", 0.0), ("MSA", "f32", "stream", 1e-5), ("MSA", "f32", "fused", 1e-5)]


def _soft_cases():
    """(case, mode) pairs that exist: the trace was recorded for the case's own check rule; toy codes have no LDS-resident shape."""
    out = []
    for path in SPA_TRACE_CASES + MSA_TRACE_CASES:
        c = load_case(path)
        if c["sumcols_trace"].shape[0] == 0:
            continue
        for alg, prec, backend, rtol in SOFT_MODES:
            if c["decoder"] != alg or (backend == "fused" and not any(k in c["code"] for k in FUSED_CODES)):
                continue
            out.append(pytest.param(path, alg, prec, backend, rtol, id="%s-%s-%s-%s" % (case_id(path), alg, prec, backend)))
    return out


@pytest.mark.parametrize("path,alg,prec,backend,rtol", _soft_cases())
def test_soft_llr_tolerance(path, alg, prec, backend, rtol):
    # marginal LLRs after 1..3 sweeps against the reference's recorded sum_cols outputs (src/bpa.py:35), through
    # ldpc_decode_soft on the streaming AND the fused (LDS-resident) kernels.
    import torch
    from ldpc_decoders_amd import bpa

    c = load_case(path)
    tr = c["sumcols_trace"]
    g, code = _code(c["code"])
    pri = _priors(c)[: tr.shape[0]]
    dt = np.float64 if prec == "f64" else np.float32
    cls = bpa.SPA if alg == "SPA" else bpa.MSA
    worst = 0.0
    for sweeps in range(1, min(tr.shape[1], c["max_iter"]) + 1):
        dec = cls(code, max_iter=sweeps, precision=prec, backend=backend)
        p_dev = torch.from_numpy(pri.astype(dt)).cuda()
        _, iters, marg = dec.handle.decode_soft_device(p_dev, None, sweeps, flags=1)  # run exactly `sweeps`
        assert dec.handle.last_stats()[0] == backend
        assert (iters.cpu().numpy() == sweeps).all()
        marg = marg.cpu().numpy().astype(np.float64)
        for f in range(tr.shape[0]):
            if c["iters"][f] < sweeps:
                continue  # upstream had already left; nothing recorded
            ref = pri[f] + tr[f, sweeps - 1]
            ok = np.isfinite(ref)
            assert ok.mean() > 0.99
            err = np.abs(marg[f][ok] - ref[ok]) / (1 + np.abs(ref[ok]))
            worst = max(worst, float(err.max()))
            assert err.max() <= rtol, (sweeps, f, err.max())
    print("soft LLR %s %s %s %s: worst |dev-ref|/(1+|ref|) = %.3g (bar %.3g)" % (case_id(path), alg, prec, backend, worst, rtol))


@pytest.mark.parametrize("backend", ["stream", "fused"])
def test_spa_fp64_special_priors(backend):
    # the fp64 sum-product rule on inputs its own tanh / log / atanh (csrc/ldpc_cn.hpp spa64_*) must treat as the reference's libm does:
    # zeros (log 0 = -inf -> NaN / 0 artefacts), saturating and infinite LLRs (tanh == 1, q == +-1 -> +-inf, inf - inf = NaN), NaN itself, tiny
    # and huge magnitudes -- marginals after one and two sweeps against the numpy oracle, NaN / inf patterns included
    import torch
    from ldpc_decoders_amd import bpa

    g, code = _code("1200_3_6_rand_ldpc_1")
    rng = np.random.RandomState(12)
    B = 48
    pri = O.biawgn_priors(-1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(2.0)), (B, g.n)), 2.0)
    specials = [0.0, -0.0, np.inf, -np.inf, 1e300, -1e300, 1e-300, 4e-320, 37.0, 38.5, 40.5, 745.0, 1e-9, 0.49999, 0.5, 0.5000001, np.nan]
    for f in range(B):
        pos = rng.choice(g.n, size=60, replace=False)
        pri[f, pos] = rng.choice(specials, size=60)
    pri[0] = 0.0
    pri[1] = np.inf
    pri[2, ::2] = -np.inf
    clean = np.ones(B, dtype=bool)  # frames whose marginals were finite-for-finite so far (a NaN-vs-inf difference, below, feeds the next sweep)
    for sweeps in (1, 2):
        dec = bpa.SPA(code, max_iter=sweeps, precision="f64", backend=backend)
        _, iters, marg = dec.handle.decode_soft_device(torch.from_numpy(pri).cuda(), None, sweeps, flags=1)
        assert dec.handle.last_stats()[0] == backend and (iters.cpu().numpy() == sweeps).all()
        marg = marg.cpu().numpy()
        v2c = pri[:, g.var].copy()
        with np.errstate(all="ignore"):
            for _ in range(sweeps):
                c2v = O.spa_check_update(g, v2c)
                ref = pri + g.sum_cols(c2v)
                v2c = ref[:, g.var] - c2v
        # finite where the reference is finite, non-finite where it is not.  WHICH non-finite value is not pinned: a row whose other edges
        # are all saturated computes exp(log t) / t, which lands on 1 (-> +-inf) or one ulp above (-> atanh = NaN) by the last bit of log and
        # exp -- numpy's, glibc's and this library's differ there (the decision is bit 0 either way for a positive sign)
        m, r = marg[clean], ref[clean]
        assert (np.isfinite(m) == np.isfinite(r)).all()
        both_inf = np.isinf(m) & np.isinf(r)
        assert (np.sign(m[both_inf]) == np.sign(r[both_inf])).all()
        ok = np.isfinite(r)
        assert ok.any() and (~ok).any()
        err = np.abs(m[ok] - r[ok]) / (1 + np.abs(r[ok]))
        assert err.max() <= 1e-9, err.max()
        clean &= ((np.isnan(marg) == np.isnan(ref)) & (np.isinf(marg) == np.isinf(ref))).all(axis=1)
    assert clean.sum() >= B // 2  # most frames agree in every NaN and inf as well


def test_soft_output_with_early_exit_matches_between_backends():
    # with the syndrome exit ON, both backends return the marginals of each frame's LAST executed sweep (0 for a frame that
    # never swept) -- fp64 min-sum: bit-identical
    import torch
    from ldpc_decoders_amd import bpa

    g, code = _code("1200_3_6_rand_ldpc_1")
    rng = np.random.RandomState(3)
    pri = O.biawgn_priors(-1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(2.2)), (300, g.n)), 2.2)
    outs = []
    for backend in ("stream", "fused"):
        dec = bpa.MSA(code, max_iter=50, precision="f64", backend=backend)
        x, it, mg = dec.handle.decode_soft_device(torch.from_numpy(pri).cuda(), None, 50)
        outs.append((x.cpu().numpy(), it.cpu().numpy(), mg.cpu().numpy()))
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()
    assert (outs[0][2] == outs[1][2]).all()
    assert ((outs[0][2] < 0) == (outs[0][0] == 1)).all()  # the hard decision is the sign of the returned marginal
    assert len(np.unique(outs[0][1])) > 5


SPA_F32_ALLOWED_FLIPS = {}  # case id -> frames whose fp32 decisions may differ from the fp64 phi statement (measured: none)


@pytest.mark.parametrize("path", [p for p in decode_cases("*_SPA_*") if "1200" in p and "bec_" not in p], ids=case_id)
def test_spa_fp32_decisions(path):
    # fp32 SPA implements the phi-domain statement of the check rule (oracle: bp_oracle.spa_phi_check_update, itself
    # tied to the reference below saturation by tests/test_oracle_golden.py::test_phi_rule_*):
    #   - against that fp64 statement: decisions identical on every frame (measured; fp32 rounding COULD flip a chaotic frame -- such a
    #     frame would have to be named in SPA_F32_ALLOWED_FLIPS)
    #   - against the reference: identical on frames whose messages stay below |LLR| = 30 upstream
    from ldpc_decoders_amd import bpa

    c = load_case(path)
    g, code = _code(c["code"])
    dec = bpa.SPA(code, max_iter=c["max_iter"], precision="f32")
    y0 = None if c["channel"] == "biawgn" else c["y"]
    pri = _priors(c)
    xhat, iters = dec.decode_batch(y0, pri.astype(np.float32))
    xo, io = O.bp_decode(g, "SPA_PHI", c["y"].astype(np.float64), pri, c["max_iter"])
    same = (xhat == xo).all(axis=1)
    O.bp_decode(g, "SPA", c["y"].astype(np.float64), pri, c["max_iter"])
    calm = np.setdiff1d(np.flatnonzero(O.bp_decode.last_peak < 30.0), c["raw_rows"])
    want = expected_xhat(c)
    calm_same = (xhat[calm] == want[calm]).all(axis=1) if len(calm) else np.ones(1, dtype=bool)
    print("fp32 sum-product, %s backend=%s: frames identical to the fp64 phi statement %d/%d (%.1f %%); to the reference on the %d frames below |LLR| 30: %.1f %%"
          % (case_id(path), dec.handle.last_stats()[0], same.sum(), len(same), 100 * same.mean(), len(calm), 100 * calm_same.mean()))
    # held to what is measured (profiles/r03H_parity_measured.txt): every frame identical to the fp64 phi statement, every calm frame
    # identical to the reference; frames allowed to differ are named here
    allowed = SPA_F32_ALLOWED_FLIPS.get(case_id(path), set())
    assert set(int(f) for f in np.flatnonzero(~same)) <= allowed
    assert (np.abs(iters - io) <= 1)[io < c["max_iter"]].all()
    if len(calm):
        assert calm_same.all()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("code_name,snr", [("1200_3_6_rand_ldpc_1", 1.0), ("1200_3_6_rand_ldpc_1", 2.0), ("1200_rho_x5_rand_ldpc_5", 2.0),
                                           ("512_3_6_rand_ldpc_2", 2.5), ("margulis", 2.0), ("7_4_hamming", 3.0)])
def test_msa_fp32_bit_exact_vs_oracle(code_name, snr, backend):
    # arbitrary fp32 priors: device fp32 min-sum == the C oracle compiled in float (same operation order)
    from ldpc_decoders_amd import bpa

    g, code = _code(code_name)
    rng = np.random.RandomState(11)
    B = 200 if g.n > 100 else 1000
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(snr)), (B, g.n))
    pri = O.biawgn_priors(y, snr).astype(np.float32)
    want_x, want_it = C.bp_decode(g, "MSA", None, pri, 50, dtype=np.float32)
    dec = bpa.MSA(code, max_iter=50, precision="f32", backend=backend)
    xhat, iters = dec.decode_batch(None, pri)
    assert (xhat == want_x).all() and (iters == want_it).all()


def test_msa_fp32_equals_fp64_reference_on_quantised_priors():
    # priors on a 2^-8 grid are exact in fp32 and every min-sum message stays exact, so the fp32 device path must
    # reproduce the fp64 *numpy* oracle (== reference) bit for bit
    from ldpc_decoders_amd import bpa

    g, code = _code("1200_3_6_rand_ldpc_1")
    rng = np.random.RandomState(5)
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(2.0)), (64, g.n))
    pri = np.round(O.biawgn_priors(y, 2.0) * 256) / 256
    xo, io = O.bp_decode(g, "MSA", y, pri, 50)
    for backend in BACKENDS:
        dec = bpa.MSA(code, max_iter=50, precision="f32", backend=backend)
        xhat, iters = dec.decode_batch(None, pri.astype(np.float32))
        assert (xhat == xo).all() and (iters == io).all()


def test_bsc_iteration0_and_ties():
    # BSC: all LLR magnitudes equal -> massive exact ties and zeros under min-sum (sgn(0)=+1, marginal==0 -> bit 0);
    # received words that already are codewords must come back untouched with iters == 0 (src/bpa.py:20,29)
    from ldpc_decoders_amd import bsc

    g, code = _code("1200_3_6_rand_ldpc_1")
    rng = np.random.RandomState(3)
    B = 96
    y = (rng.random_sample((B, g.n)) < 0.035).astype(np.int64)
    y[:7] = 0
    y[7] = 1  # all-ones is a codeword of a (3,6)-regular code
    for alg in ("MSA", "SPA"):
        dec = getattr(bsc, alg)(0.035, code, max_iter=30, precision="f64")
        xhat, iters = dec.decode_batch(y)
        xo, io = O.bp_decode(g, alg, y, O.bsc_priors(y, 0.035), 30)
        if alg == "MSA":
            assert (xhat == xo).all() and (iters == io).all()
        assert (iters[:8] == 0).all() and (xhat[:8] == y[:8]).all()
    one = bsc.MSA(0.035, code, max_iter=30).decode(y[0])
    assert one is not None and (one == y[0]).all()


def test_ragged_and_edge_batches():
    from ldpc_decoders_amd import bpa

    g, code = _code("512_3_6_rand_ldpc_2")
    rng = np.random.RandomState(9)
    for B in (1, 63, 64, 65, 130):
        y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(2.5)), (B, g.n))
        pri = O.biawgn_priors(y, 2.5)
        for backend in BACKENDS:
            xhat, iters = bpa.MSA(code, max_iter=20, precision="f64", backend=backend).decode_batch(None, pri)
            xo, io = C.bp_decode(g, "MSA", None, pri, 20)
            assert (xhat == xo).all() and (iters == io).all()
    # max_iter = 1 and the "no early exit" flag
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(2.5)), (70, g.n))
    pri = O.biawgn_priors(y, 2.5)
    dec = bpa.MSA(code, max_iter=1, precision="f64")
    xhat, iters = dec.decode_batch(None, pri)
    xo, io = C.bp_decode(g, "MSA", None, pri, 1)
    assert (xhat == xo).all() and (iters == 1).all()
    with pytest.raises(ValueError):
        dec.decode_batch(None, pri[:, :-1])


@pytest.mark.parametrize("nw", ["1", "2", "4"])
def test_fused_variants_ragged_and_iteration0(nw, monkeypatch):
    # both fused kernels (1 and 2 wavefronts per frame) on tiny / ragged batches, with and without the iteration-0 word
    from ldpc_decoders_amd import bpa, bsc

    monkeypatch.setenv("LDPC_FUSED_NW", nw)
    g, code = _code("1200_3_6_rand_ldpc_1")
    rng = np.random.RandomState(21)
    for B in (1, 5, 130):
        y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(2.2)), (B, g.n))
        pri = O.biawgn_priors(y, 2.2).astype(np.float32)
        dec = bpa.MSA(code, max_iter=40, precision="f32", backend="fused")
        assert dec.handle.fused_info()["waves_per_frame"] == float(nw)
        xhat, iters = dec.decode_batch(None, pri)
        xo, io = C.bp_decode(g, "MSA", None, pri, 40, dtype=np.float32)
        assert (xhat == xo).all() and (iters == io).all()
    yb = (rng.random_sample((150, g.n)) < 0.03).astype(np.int64)
    yb[:5] = 0
    yb[5] = 1
    dec = bsc.MSA(0.03, code, max_iter=30, precision="f32", backend="fused")
    xhat, iters = dec.decode_batch(yb)
    pri = O.bsc_priors(yb, 0.03).astype(np.float32)
    xo, io = C.bp_decode(g, "MSA", yb.astype(np.float32), pri, 30, dtype=np.float32)
    assert (xhat == xo).all() and (iters == io).all() and (iters[:6] == 0).all()
    # max_iter = 1 and the no-early-exit flag
    import torch
    h = bpa.MSA(code, max_iter=3, precision="f32", backend="fused").handle
    x3, i3 = h.decode_device(torch.from_numpy(pri).cuda(), None, 3, flags=1)
    assert (i3.cpu().numpy() == 3).all()


@pytest.mark.parametrize("nw", ["2", "4"])
def test_fused_erasure_decoder_variants(nw, monkeypatch):
    # bit-sliced erasure decoder on the LDS (2 and 4 wavefronts per slab of 32 frames) against the C oracle incl. stopping sets, max_iter
    # cuts, ragged last slabs (1500 = 46 slabs + 28 frames) and the fused simulate counters (channel + decode + count in one kernel) --
    # repeated to shake out hand-off races
    import torch
    from ldpc_decoders_amd._device import DecoderHandle

    monkeypatch.setenv("LDPC_FUSED_NW", nw)
    g, code = _code("1200_3_6_rand_ldpc_1")
    h = DecoderHandle(code, "BEC", "f32", "fused")
    assert h.fused_info()["waves_per_frame"] == float(nw)
    for rep in range(3):
        _, y = h.channel_device("bec", 0.41, rep & 1, 77 + rep, 2, 5000 * rep, 1500)
        yh = y.cpu().numpy()
        for mi in (50, 2):
            xh, it = h.decode_device(None, y, mi)
            xo, io = C.bec_decode(g, yh, mi)
            assert (xh.cpu().numpy() == xo).all() and (it.cpu().numpy() == io).all()
        cnt = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
        h.simulate("bec", 0.41, rep & 1, 77 + rep, 2, 5000 * rep, 1500, 50, cnt, hist_bins=51)
        xo, io = C.bec_decode(g, yh, 50)
        err = (xo != (rep & 1)).sum(axis=1)
        c = cnt.cpu().numpy()
        assert (c[0], c[1], c[2], c[3]) == (1500, (err > 0).sum(), err.sum(), io.sum())
        assert (c[4:] == np.bincount(np.minimum(io, 50), minlength=51)).all()


@pytest.mark.parametrize("nw,alg", [("1", "MSA"), ("2", "MSA"), ("2", "BEC")])
def test_fused_irregular_shapes(nw, alg, monkeypatch):
    # the irregular n = 1200 code on both of its fused shapes (one wave per frame; two waves with the system row), ragged batches; the
    # bit-sliced erasure decoder has the two-wave shape
    from ldpc_decoders_amd import bec, bpa

    monkeypatch.setenv("LDPC_FUSED_NW", nw)
    g, code = _code("1200_rho_x5_rand_ldpc_5")
    rng = np.random.RandomState(31)
    for B in (3, 257):
        if alg == "MSA":
            y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(1.9)), (B, g.n))
            pri = O.biawgn_priors(y, 1.9).astype(np.float32)
            dec = bpa.MSA(code, max_iter=40, precision="f32", backend="fused")
            xhat, iters = dec.decode_batch(None, pri)
            xo, io = C.bp_decode(g, "MSA", None, pri, 40, dtype=np.float32)
        else:
            ye = (rng.random_sample((B, g.n)) < 0.42).astype(np.int64) * 2
            dec = bec.SPA(0.42, code, max_iter=40, backend="fused")
            xhat, iters = dec.decode_batch(ye)
            xo, io = C.bec_decode(g, ye, 40)
        assert dec.handle.fused_info()["waves_per_frame"] == float(nw)
        assert (xhat == xo).all() and (iters == io).all()


def test_layout_plan_store(monkeypatch, tmp_path):
    # the LDS placement only moves data around: shipped plan (ldpc_decoders_amd/plans), a plan annealed now, a plan saved and
    # read back, and the trivial placement all give the same bits as the oracle
    from ldpc_decoders_amd import bpa

    g, code = _code("1200_3_6_rand_ldpc_1")
    rng = np.random.RandomState(5)
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(1.8)), (200, g.n))
    pri = O.biawgn_priors(y, 1.8).astype(np.float32)
    xo, io = C.bp_decode(g, "MSA", None, pri, 50, dtype=np.float32)

    def run():
        dec = bpa.MSA(code, max_iter=50, precision="f32", backend="fused")
        xhat, iters = dec.decode_batch(None, pri)
        assert (xhat == xo).all() and (iters == io).all()
        return dec.handle.fused_info()["conflict_cycles_planned"]

    shipped = run()
    monkeypatch.setenv("LDPC_FUSED_LAYOUT", "replan")
    monkeypatch.setenv("LDPC_FUSED_PLAN_MOVES", "400000")
    monkeypatch.setenv("LDPC_FUSED_PLAN_SAVE", str(tmp_path))
    fresh = run()
    assert len(os.listdir(str(tmp_path))) == 1 and shipped < 60 < fresh < 400
    monkeypatch.delenv("LDPC_FUSED_LAYOUT")
    monkeypatch.delenv("LDPC_FUSED_PLAN_SAVE")
    monkeypatch.setenv("LDPC_FUSED_PLAN_DIR", str(tmp_path))  # searched before the package's own plans
    assert run() == fresh
    monkeypatch.setenv("LDPC_FUSED_LAYOUT", "identity")
    assert run() > 400


@pytest.mark.parametrize("code_name,alg", [("1200_3_6_rand_ldpc_1", "MSA"), ("1200_rho_x5_rand_ldpc_5", "MSA"), ("margulis", "MSA"),
                                           ("1200_3_6_rand_ldpc_1", "BEC"), ("margulis", "SPA")])
def test_fused_multiwave_kernels_are_deterministic(code_name, alg):
    # the waves of a frame hand verdicts to each other through LDS words: repeated decodes of one resident batch must give
    # identical outputs (a lost or stale hand-off shows up as a differing iteration count)
    import torch
    from ldpc_decoders_amd._device import DecoderHandle

    g, code = _code(code_name)
    h = DecoderHandle(code, alg, "f32", "fused")
    assert h.fused_info()["waves_per_frame"] >= 2
    if alg == "BEC":
        pri, y = h.channel_device("bec", 0.41, 0, 5, 0, 0, 6000)
    else:
        pri, y = h.channel_device("biawgn", 1.8, 0, 5, 0, 0, 6000)
        y = None
    x0, i0 = h.decode_device(pri, y, 40)
    x0, i0 = x0.clone(), i0.clone()
    assert len(torch.unique(i0)) > 3
    for _ in range(6):
        x1, i1 = h.decode_device(pri, y, 40)
        assert torch.equal(x1, x0) and torch.equal(i1, i0)


def test_fp64_min_sum_on_the_lds_kernel():
    # the fp64 LDS kernel (the reference's own arithmetic) against the fp64 C oracle: ragged batches, BSC iteration-0 exits,
    # max_iter cuts, no-early-exit flag; and it is the kernel "auto" picks for regular codes in fp64
    import torch
    from ldpc_decoders_amd import bpa, bsc

    for name in ("1200_3_6_rand_ldpc_1", "512_3_6_rand_ldpc_2", "1200_rho_x5_rand_ldpc_5", "margulis"):  # regular, small, irregular, 96 KB frame
        g, code = _code(name)
        rng = np.random.RandomState(17)
        for B, mi in ((1, 50), (67, 50), (400, 7)):
            y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(2.1)), (B, g.n))
            pri = O.biawgn_priors(y, 2.1)
            dec = bpa.MSA(code, max_iter=mi, precision="f64")
            xhat, iters = dec.decode_batch(None, pri)
            assert dec.handle.last_stats()[0] == "fused" and dec.handle.fused_info()["lds_bytes_per_frame"] > 40000
            xo, io = C.bp_decode(g, "MSA", None, pri, mi, dtype=np.float64)
            assert (xhat == xo).all() and (iters == io).all()
    g, code = _code("1200_3_6_rand_ldpc_1")
    yb = (np.random.RandomState(3).random_sample((200, g.n)) < 0.035).astype(np.int64)
    yb[:4] = 0
    dec = bsc.MSA(0.035, code, max_iter=30, precision="f64")
    xhat, iters = dec.decode_batch(yb)
    xo, io = C.bp_decode(g, "MSA", yb.astype(np.float64), O.bsc_priors(yb, 0.035), 30, dtype=np.float64)
    assert dec.dec.handle.last_stats()[0] == "fused" and (xhat == xo).all() and (iters == io).all() and (iters[:4] == 0).all()
    h = bpa.MSA(code, max_iter=3, precision="f64", backend="fused").handle
    x3, i3 = h.decode_device(torch.from_numpy(O.bsc_priors(yb, 0.035)).cuda(), None, 3, flags=1)
    assert (i3.cpu().numpy() == 3).all()


@pytest.mark.parametrize("code_name", ["1200_3_6_rand_ldpc_1", "512_3_6_rand_ldpc_2", "1200_3_6_ldpc"])
def test_fp64_two_and_four_waves_per_frame_agree(code_name, monkeypatch):
    # the four-wave fp64 min-sum shape (10 check rows, last wave short: csrc fused_check_rows) against its two-wave sibling and the C oracle:
    # decisions and iteration counts of ragged batches, then the in-kernel Monte-Carlo counters (same Philox frames)
    import torch
    from ldpc_decoders_amd import bpa
    from ldpc_decoders_amd._device import DecoderHandle

    g, code = _code(code_name)
    rng = np.random.RandomState(5)
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(1.8)), (333, g.n))
    pri = O.biawgn_priors(y, 1.8)
    xo, io = C.bp_decode(g, "MSA", None, pri, 50, dtype=np.float64)
    out, counters = {}, {}
    for nw in ("2", "4"):
        monkeypatch.setenv("LDPC_FUSED_NW", nw)
        dec = bpa.MSA(code, max_iter=50, precision="f64", backend="fused")
        assert dec.handle.fused_info()["waves_per_frame"] == float(nw)
        out[nw] = dec.decode_batch(None, pri)
        assert (out[nw][0] == xo).all() and (out[nw][1] == io).all()
        h = DecoderHandle(code, "MSA", "f64", "fused")
        cnt = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
        h.simulate("biawgn", 1.5, 0, 11, 0, 0, 20000, 50, cnt, hist_bins=51)
        counters[nw] = cnt.cpu().numpy()
    monkeypatch.delenv("LDPC_FUSED_NW")
    assert (counters["2"] == counters["4"]).all() and counters["4"][0] == 20000 and counters["4"][1] > 0
    assert bpa.MSA(code, max_iter=50, precision="f64").handle.fused_info()["waves_per_frame"] == 4.0  # what "auto" picks


@pytest.mark.parametrize("code_name", ["1200_3_6_rand_ldpc_1", "1200_rho_x5_rand_ldpc_5", "margulis"])
def test_fp64_lds_kernel_is_deterministic(code_name):
    # repeated decodes of one resident batch on the multi-wave fp64 kernels (2 and 8 waves per frame): identical outputs
    import torch
    from ldpc_decoders_amd._device import DecoderHandle

    g, code = _code(code_name)
    h = DecoderHandle(code, "MSA", "f64", "fused")
    assert h.fused_info()["waves_per_frame"] >= 2
    pri, _ = h.channel_device("biawgn", 1.9, 0, 5, 0, 0, 5000)
    x0, i0 = h.decode_device(pri, None, 40)
    x0, i0 = x0.clone(), i0.clone()
    assert len(torch.unique(i0)) > 3
    for _ in range(5):
        x1, i1 = h.decode_device(pri, None, 40)
        assert torch.equal(x1, x0) and torch.equal(i1, i0)
