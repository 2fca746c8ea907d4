"""Per-frame early termination on the streaming backend (src/bpa.py:28-29: a frame that has left costs nothing): live frames are
re-gathered into dense tiles while a batch thins out.  The repack only moves state, so every result must stay bit-identical --
checked against the C oracle, against a run with the repack switched off, and against the LDS-resident backend."""
import numpy as np
import pytest

import bp_oracle as O
import c_oracle as C
from helpers import golden_edges

pytestmark = pytest.mark.gpu


def _priors(seed, B, n, snr):
    rng = np.random.RandomState(seed)
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(snr)), (B, n))
    return O.biawgn_priors(y, snr)


@pytest.mark.parametrize("fold", ["1", "0"], ids=["folded", "separate"])
@pytest.mark.parametrize("prec,dt", [("f64", np.float64), ("f32", np.float32)])
@pytest.mark.parametrize("alg", ["MSA", "SPA"])
def test_repack_is_bit_transparent(monkeypatch, prec, dt, alg, fold):
    # both forms of the repack: folded into the sweep behind it (k_repack_map + the GATHER passes, the default where those passes are built)
    # and the separate copy kernel (k_repack: LDPC_STREAM_REPACK_FOLD=0, and every code with node degrees beyond 8)
    from ldpc_decoders_amd import bpa
    from ldpc_decoders_amd.codes import Code

    monkeypatch.setenv("LDPC_STREAM_REPACK_FOLD", fold)
    g = golden_edges("1200_3_6_rand_ldpc_1")
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    B = 64 * 40 + 17  # ragged last tile
    pri = _priors(11, B, g.n, 2.2).astype(dt)  # frames leave after 5..50 sweeps, a few never do
    cls = bpa.MSA if alg == "MSA" else bpa.SPA
    monkeypatch.setenv("LDPC_STREAM_REPACK", "0")
    ref = cls(code, max_iter=50, precision=prec, backend="stream")
    x0, i0 = ref.decode_batch(None, pri)
    assert ref.handle.last_repacks() == 0
    monkeypatch.setenv("LDPC_STREAM_REPACK", "1")
    monkeypatch.setenv("LDPC_STREAM_REPACK_FILL", "0.97")  # eager: several repacks in one decode
    dec = cls(code, max_iter=50, precision=prec, backend="stream")
    x1, i1 = dec.decode_batch(None, pri)
    assert dec.handle.last_stats()[0] == "stream" and dec.handle.last_repacks() >= 2
    assert (x1 == x0).all() and (i1 == i0).all()
    assert (i1 < 50).any() and (i1 == 50).any() and len(np.unique(i1)) > 8
    if alg == "MSA":  # min-sum is exact arithmetic: the C oracle in the same precision gives the same bits
        xo, io = C.bp_decode(g, "MSA", None, pri, 50, dtype=dt)
        assert (x1 == xo).all() and (i1 == io).all()
    monkeypatch.delenv("LDPC_STREAM_REPACK_FILL")
    dflt = cls(code, max_iter=50, precision=prec, backend="stream")
    x2, i2 = dflt.decode_batch(None, pri)
    assert (x2 == x0).all() and (i2 == i0).all()


@pytest.mark.parametrize("alg", ["MSA", "SPA"])
@pytest.mark.parametrize("name,B", [("1200_3_6_rand_ldpc_1", 64 * 41 + 17), ("1200_rho_x5_rand_ldpc_5", 64 * 12 + 3)])
def test_repack_of_the_fp16_storage_mode_is_bit_transparent(monkeypatch, alg, name, B):
    # the fp16 storage mode works on pair-tiles of 128 frames; its repack draws the two halves of a destination lane from two unrelated
    # (tile, lane) sources.  Moving state must not change a single decision or iteration count (odd tile counts, a ragged last tile,
    # irregular degrees, several repacks in one decode, and the Monte-Carlo entry point).
    import torch
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges(name)
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    pri = torch.from_numpy(_priors(5, B, g.n, 2.3).astype(np.float32)).cuda()
    h = DecoderHandle(code, alg, "f16")
    monkeypatch.setenv("LDPC_STREAM_REPACK", "0")
    x0, i0 = h.decode_device(pri, None, 50)
    assert h.last_stats()[0] == "stream" and h.last_repacks() == 0
    cnt0 = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
    h.simulate("biawgn", 2.3, 0, 9, 1, 0, B, 50, cnt0, hist_bins=51)
    monkeypatch.setenv("LDPC_STREAM_REPACK", "1")
    monkeypatch.setenv("LDPC_STREAM_REPACK_FILL", "0.97")  # eager: several repacks in one decode
    x1, i1 = h.decode_device(pri, None, 50)
    assert h.last_repacks() >= 2
    assert (x1 == x0).all() and (i1 == i0).all()
    assert 0 < (i0.cpu().numpy() < 50).mean()  # frames did leave early
    cnt1 = torch.zeros_like(cnt0)
    h.simulate("biawgn", 2.3, 0, 9, 1, 0, B, 50, cnt1, hist_bins=51)
    assert h.last_repacks() >= 1 and (cnt1 == cnt0).all()


def test_repack_with_received_word(monkeypatch):
    # BSC: the iteration-0 exit of the received word (src/bpa.py:20,29) happens before any repack
    from ldpc_decoders_amd import bpa
    from ldpc_decoders_amd.codes import Code

    g = golden_edges("1200_3_6_rand_ldpc_1")
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    rng = np.random.RandomState(3)
    B, p = 64 * 24, 0.02
    y = (rng.random_sample((B, g.n)) < p).astype(np.uint8)
    y[:5] = 0  # frames that leave at iteration 0
    pri = (np.log((1 - p) / p) * (1 - 2.0 * y)).astype(np.float64)
    monkeypatch.setenv("LDPC_STREAM_REPACK_FILL", "0.97")
    dec = bpa.MSA(code, max_iter=200, precision="f64", backend="stream")
    x1, i1 = dec.decode_batch(y, pri)
    xo, io = C.bp_decode(g, "MSA", y, pri, 200, dtype=np.float64)
    assert dec.handle.last_repacks() >= 1
    assert (i1[:5] == 0).all() and (x1 == xo).all() and (i1 == io).all()


@pytest.mark.parametrize("name", ["1200_rho_x5_rand_ldpc_5", "margulis"])
def test_folded_repack_on_irregular_and_large_codes(monkeypatch, name):
    # GATHER passes of the other instantiations: irregular degrees (dc <= 6, dv <= 8: no fixed row / column length) and the n = 2640 code,
    # fp32 and fp64, several repacks per decode; against the decode without any repack and (min-sum) the C oracle
    from ldpc_decoders_amd import bpa
    from ldpc_decoders_amd.codes import Code

    g = golden_edges(name)
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    B = 64 * 13 + 5
    for prec, dt in (("f32", np.float32), ("f64", np.float64)):
        pri = _priors(23, B, g.n, 2.1).astype(dt)
        monkeypatch.setenv("LDPC_STREAM_REPACK", "0")
        x0, i0 = bpa.MSA(code, max_iter=40, precision=prec, backend="stream").decode_batch(None, pri)
        monkeypatch.setenv("LDPC_STREAM_REPACK", "1")
        monkeypatch.setenv("LDPC_STREAM_REPACK_FILL", "0.97")
        dec = bpa.MSA(code, max_iter=40, precision=prec, backend="stream")
        x1, i1 = dec.decode_batch(None, pri)
        assert dec.handle.last_repacks() >= 1 and (x1 == x0).all() and (i1 == i0).all() and len(np.unique(i1)) > 4
        xo, io = C.bp_decode(g, "MSA", None, pri, 40, dtype=dt)
        assert (x1 == xo).all() and (i1 == io).all()


def test_repack_large_code_mid_snr():
    # the (3,6) n = 64 800 shape of BASELINE config 5 at an SNR where every frame converges after a different number of sweeps:
    # repacked streaming decode == C oracle on every frame, iteration counts included
    from ldpc_decoders_amd import bpa, codes

    code = codes.rand_reg_ldpc(64800, 3, 6, np.random.RandomState(20261002))
    B = 64 * 6
    pri = _priors(5, B, code.n, 2.0).astype(np.float32)  # frames leave after 12..22 sweeps
    dec = bpa.MSA(code, max_iter=60, precision="f32", backend="stream")
    x1, i1 = dec.decode_batch(None, pri)

    class G:
        m, n, chk, var = code.m, code.n, code.edge_chk, code.edge_var

    xo, io = C.bp_decode(G, "MSA", None, pri, 60, dtype=np.float32)  # every frame
    assert (x1 == xo).all() and (i1 == io).all()
    assert dec.handle.last_repacks() >= 1 and i1.min() < i1.max()
