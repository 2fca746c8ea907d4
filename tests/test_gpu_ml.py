"""GPU parity tests of the ML decoder kernels (ldpc_ml_decode / ldpc_ml_simulate) through the C ABI, against vectors
captured from the reference's ML classes and against the oracle (oracle/ml_oracle.py)."""
import json
import os

import numpy as np
import pytest

import bp_oracle as O
import ml_oracle as M
from helpers import GOLDEN
from test_oracle_ml import ml_arrays, ml_cases

pytestmark = pytest.mark.gpu


def tie_bits(mask, K):
    m = mask.cpu().numpy().view(np.uint32)
    return ((m[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(m.shape[0], -1)[:, :K].astype(bool)


@pytest.mark.parametrize("case", ml_cases(), ids=lambda c: "%s-%s-%s" % (c["channel"], c["code"], c["param"]))
def test_metric_and_tie_sets_bit_exact_vs_reference(case):
    import torch

    from ldpc_decoders_amd._device import MlHandle

    a = ml_arrays(case)
    ch = case["channel"]
    h = MlHandle(a["cb"], ch, "f64")
    y = torch.from_numpy(np.ascontiguousarray(a["y"], dtype=np.float64 if ch == "biawgn" else np.uint8)).cuda()
    coef = M.ml_coefficients(ch, case["param"])
    out = h.decode_device(y, coef)
    lp = a["log_prob"]
    with np.errstate(all="ignore"):
        want_best = lp.max(axis=1)
        want_ties = lp == want_best[:, None]
    assert np.array_equal(out["best"].cpu().numpy(), want_best, equal_nan=True)  # fp64, bit for bit
    got_ties = tie_bits(out["tie_mask"], h.K)
    nan_rows = np.isnan(want_best)  # eps = 1 style degenerate rows: the reference itself fails there (empty tie set)
    assert np.array_equal(got_ties[~nan_rows], want_ties[~nan_rows])
    assert np.array_equal(out["ties"].cpu().numpy()[~nan_rows], want_ties.sum(axis=1)[~nan_rows])
    # without draws: the first maximiser
    assert np.array_equal(out["index"].cpu().numpy()[~nan_rows], want_ties.argmax(axis=1)[~nan_rows])
    # with draws: maximiser number floor(draw * ties / 2^32)
    draws = np.random.default_rng(3).integers(0, 2 ** 32, size=len(lp), dtype=np.uint64)
    pick = torch.from_numpy(draws.astype(np.uint32).view(np.int32)).cuda()
    out2 = h.decode_device(y, coef, pick=pick, want_mask=False)
    idx = out2["index"].cpu().numpy()
    for f in np.flatnonzero(~nan_rows):
        s = np.flatnonzero(want_ties[f])
        assert idx[f] == s[(int(draws[f]) * len(s)) >> 32]
    assert np.array_equal(out2["xhat"].cpu().numpy()[~nan_rows], a["cb"][idx[~nan_rows]].astype(np.uint8))


@pytest.mark.parametrize("case", ml_cases(), ids=lambda c: "%s-%s-%s" % (c["channel"], c["code"], c["param"]))
def test_registry_ml_replays_reference_sequence(case, monkeypatch):
    # reference: np.random.seed(s); per frame  x -> Channel.send -> ML.decode  (the pick consumes the same global stream)
    from ldpc_decoders_amd import codes
    from ldpc_decoders_amd.models import models

    a = ml_arrays(case)
    mod = models[case["channel"]]
    code = codes.get_code(case["code"])
    chan, dec = mod.Channel(case["param"]), mod.ML(case["param"], code, max_iter=10)
    assert dec.id_keys == []
    np.random.seed(case["seed"])
    for f in range(min(case["frames"], 120)):
        x = code.cb[np.random.choice(code.cb.shape[0], 1)[0]] if case["codeword"] == -1 else np.zeros(code.get_n(), dtype=np.int64) + case["codeword"]
        xh = dec.decode(chan.send(x))
        assert np.array_equal(xh, a["xhat"][f]), "frame %d" % f


def test_ml_kats_on_device():
    from ldpc_decoders_amd import codes
    from ldpc_decoders_amd.models import models

    with open(os.path.join(GOLDEN, "ml_kat.json")) as fp:
        kats = json.load(fp)
    for k in kats:
        np.random.seed(k["np_seed"])
        est = models[k["channel"]].ML(k["param"], codes.get_code(k["code"]), max_iter=100).decode(np.array(k["received"]))
        assert list(est) == k["reference_estimate"]


def _ml_main_cases():
    with open(os.path.join(GOLDEN, "main_counters_ml.json")) as fp:
        return json.load(fp)


@pytest.mark.parametrize("run", _ml_main_cases(), ids=lambda r: r["argline"].replace(" ", "_")[:60])
def test_exact_mode_reproduces_reference_ml_counters(run, tmp_path):
    from ldpc_decoders_amd import main

    argv = run["argline"].split() + ["--data_dir", str(tmp_path), "--console", "--exact", "--np-seed", str(run["seed"])]
    main.main(argv)
    with open(os.path.join(str(tmp_path), run["file_name"])) as fp:
        got = json.load(fp)
    want = run["result"]
    assert list(got) == list(want)
    for key in ("tot", "wec", "bec"):
        assert got[key] == want[key]


@pytest.mark.parametrize("channel,code,param,codeword", [("biawgn", "7_4_hamming", 2.0, 0), ("biawgn", "12_3_4_ldpc", 1.0, 0),
                                                         ("bsc", "7_4_hamming", 0.1, 1), ("bsc", "12_3_4_ldpc", 0.15, 0),
                                                         ("bec", "7_4_hamming", 0.3, 0), ("bec", "6_2_3_ldpc", 0.5, 0)])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_ml_simulate_equals_channel_decode_count(channel, code, param, codeword, precision):
    """ldpc_ml_simulate == ldpc_channel (same Philox keys) -> oracle ML with the device's tie-break word -> host count;
    and its counters do not depend on how the frame range is cut."""
    import torch

    from ldpc_decoders_amd import _lib, codes
    from ldpc_decoders_amd._device import MlHandle

    c = codes.get_code(code)
    cb = c.cb
    n, B, seed, stream, frame0 = c.get_n(), 3000, 0x5EED, 2, 12345
    h = MlHandle(cb, channel, precision)
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    h.simulate(channel, param, codeword, seed, stream, frame0, B, 0, cnt)
    cnt2 = torch.zeros(4, dtype=torch.int64, device="cuda")
    for a, b in ((0, 1), (1, 1000), (1000, B)):
        h.simulate(channel, param, codeword, seed, stream, frame0 + a, b - a, 0, cnt2)
    assert torch.equal(cnt, cnt2)
    # the same frames through the stand-alone channel kernel
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    if channel == "biawgn":
        y = torch.empty((B, n), dtype=torch.float64 if precision == "f64" else torch.float32, device="cuda")
        _lib.check(lib.ldpc_channel(_lib.CHANNEL[channel] | _lib.CH_RAW_OBSERVATION, _lib.DTYPE[precision], param, codeword, seed, stream,
                                    frame0, B, n, y.data_ptr(), None, st))
        yh = y.cpu().numpy().astype(np.float64)
        # raw observations are the noise model itself (up to the transcendental units' accuracy)
        z = np.stack([O.device_biawgn_noise(seed, stream, frame0 + f, n) for f in range(50)])
        assert np.allclose(yh[:50], (2 * codeword - 1) + np.sqrt(10 ** (-param / 10)) * z, atol=2e-4 if precision == "f32" else 1e-8)
    else:
        y = torch.empty((B, n), dtype=torch.uint8, device="cuda")
        _lib.check(lib.ldpc_channel(_lib.CHANNEL[channel], 0, param, codeword, seed, stream, frame0, B, n, None, y.data_ptr(), st))
        yh = y.cpu().numpy().astype(np.int64)
    coef = M.ml_coefficients(channel, param)
    x = np.zeros(n, dtype=np.int64) + codeword
    tot = wec = bec = 0
    with np.errstate(all="ignore"):
        for f in range(B):
            s = M.ml_tie_set(M.ml_log_prob(channel, cb, yh[f], coef))
            ctr = np.array([[0xFFFFFFFF, stream, (frame0 + f) & 0xFFFFFFFF, (frame0 + f) >> 32]], dtype=np.uint32)
            key = np.array([[seed & 0xFFFFFFFF, seed >> 32]], dtype=np.uint32)
            draw = int(O.philox4x32(ctr, key)[0, 0])
            e = int((cb[s[(draw * len(s)) >> 32]] != x).sum())
            tot, wec, bec = tot + 1, wec + (e > 0), bec + e
    assert cnt.cpu().tolist()[:3] == [tot, wec, bec]


def test_ml_device_mode_cli(tmp_path):
    """Device-noise Monte-Carlo with decoder ML writes the reference's schema; rates agree with the reference's own
    published point for this code (data/output/biawgn-7_4_hamming-ML-10-1.json: WER 0.083 at 2 dB from ~100 word errors; 0.113 in
    the seeded golden run of 60 word errors)."""
    from ldpc_decoders_amd import main

    res = main.main("biawgn 7_4_hamming ML --codeword 0 --min-wec 2000 --params 2.0 --batch 16384".split() + ["--data_dir", str(tmp_path), "--console"])
    data = json.load(open(os.path.join(str(tmp_path), "biawgn-7_4_hamming-ML-0-2000.json")))
    assert list(data)[:5] == ["channel", "code", "decoder", "codeword", "min_wec"]
    assert 0.07 < data["wer"]["2.0"] < 0.135 and res[2.0]["tot"] % 16384 == 0


@pytest.mark.parametrize("channel,param", [("biawgn", 1.0), ("bsc", 0.2), ("bec", 0.5)])
def test_ml_larger_codebook_vs_oracle(channel, param):
    # a codebook beyond one 32-bit tie word (k = 7 -> 128 words, n = 20 >= 8: numpy's 8-accumulator summation branch), ragged batch
    import torch

    from ldpc_decoders_amd._device import MlHandle
    from ldpc_decoders_amd.models import models

    rng = np.random.RandomState(9)
    G = rng.randint(0, 2, (7, 20))
    msgs = ((np.arange(128)[:, None] >> np.arange(7)) & 1)
    cb = (msgs @ G) % 2
    h = MlHandle(cb, channel, "f64")
    assert h.W == 4
    B = 333
    np.random.seed(12)
    chan = models[channel].Channel(param)
    Y = np.stack([chan.send(cb[rng.randint(128)]) for _ in range(B)])
    y = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float64 if channel == "biawgn" else np.uint8)).cuda()
    coef = M.ml_coefficients(channel, param)
    out = h.decode_device(y, coef)
    ties = tie_bits(out["tie_mask"], 128)
    best = out["best"].cpu().numpy()
    with np.errstate(all="ignore"):
        for f in range(B):
            lp = M.ml_log_prob(channel, cb, Y[f], coef)
            assert best[f] == lp.max() and np.array_equal(np.flatnonzero(ties[f]), M.ml_tie_set(lp))
    assert np.array_equal(out["index"].cpu().numpy(), ties.argmax(axis=1)) and np.array_equal(out["ties"].cpu().numpy(), ties.sum(axis=1))
