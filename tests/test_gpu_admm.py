"""GPU parity tests of the ADMM LP decoder (ldpc_admm_decode through the C ABI): bit-identical estimates and iteration
counts against vectors captured from the reference's ADMM class, against the C oracle on larger batches, and the
reference's main.py counters in --exact mode."""
import json
import os

import numpy as np
import pytest

import admm_oracle as A
from helpers import CODES_DIR, GOLDEN
from test_oracle_admm import admm_arrays, admm_cases, graph_of

pytestmark = pytest.mark.gpu


def _code(name):
    from ldpc_decoders_amd import codes

    os.environ.setdefault(codes.file_codes_dir_string, CODES_DIR)
    return codes.get_code(name)


@pytest.mark.parametrize("case", admm_cases(), ids=lambda c: "%s-%s-%s" % (c["channel"], c["code"], c["param"]))
def test_admm_bit_exact_vs_reference(case, monkeypatch):
    from ldpc_decoders_amd import admm, codes

    monkeypatch.setenv(codes.file_codes_dir_string, CODES_DIR)
    a = admm_arrays(case)
    dec = admm.ADMM(codes.get_code(case["code"]), mu=case["mu"], eps=case["eps"], max_iter=case["max_iter"],
                    allow_pseudo=case["allow_pseudo"], log_freq=5.0)  # unknown keywords are ignored, as upstream
    est, iters = dec.decode_batch(a["gamma"])
    assert np.array_equal(iters, a["iters"])
    assert np.array_equal(np.asarray(est, dtype=np.float64), a["xhat"], equal_nan=True)
    hist = np.bincount(a["iters"], minlength=2000)
    assert dec.stats()["iter"] == hist.tolist() and dec.stats()["average"] == pytest.approx(a["iters"].mean())


@pytest.mark.parametrize("case", [c for c in admm_cases() if c["code"] in ("7_4_hamming", "12_3_4_ldpc", "4_2_test")],
                         ids=lambda c: "%s-%s-%s" % (c["channel"], c["code"], c["param"]))
def test_registry_admm_replays_reference_sequence(case):
    # reference: np.random.seed(s); per frame  Channel.send -> ADMM.decode  through models[channel]
    from ldpc_decoders_amd import codes
    from ldpc_decoders_amd.models import models

    a = admm_arrays(case)
    mod = models[case["channel"]]
    code = codes.get_code(case["code"])
    chan = mod.Channel(case["param"])
    dec = mod.ADMM(case["param"], code, mu=case["mu"], eps=case["eps"], max_iter=case["max_iter"], allow_pseudo=case["allow_pseudo"])
    assert dec.id_keys == ["mu", "eps", "max_iter", "allow_pseudo"]
    x = np.zeros(code.get_n(), dtype=np.int64) + case["codeword"]
    np.random.seed(case["seed"])
    for f in range(60):
        y = chan.send(x)
        assert np.array_equal(np.asarray(y, dtype=np.float64), a["y"][f])
        assert np.array_equal(np.asarray(dec.decode(y), dtype=np.float64), a["xhat"][f], equal_nan=True)


def test_admm_batches_vs_c_oracle():
    # ragged batch sizes, frames leaving at different iterations, max_iter <= 0 (no cap) on a small code
    import torch
    from ldpc_decoders_amd._device import AdmmHandle

    rng = np.random.RandomState(4)
    for name, B, snr, max_iter in (("1200_3_6_rand_ldpc_1", 130, 2.4, 150), ("512_3_6_rand_ldpc_2", 70, 2.6, 60), ("7_4_hamming", 1000, 2.0, -1),
                                   ("1200_rho_x5_rand_ldpc_5", 40, 2.0, 120)):
        code = _code(name)
        g = graph_of(name)
        gamma = -2 * (-1 + rng.normal(0, np.sqrt(10 ** (-snr / 10)), (B, code.n))) / 10 ** (-snr / 10)
        xo, io, co = A.admm_decode(g, gamma, 3.0, 1e-5, max_iter)
        h = AdmmHandle(code)
        x, it, cv = h.decode_device(torch.from_numpy(gamma).cuda(), 3.0, 1e-5, max_iter)
        assert np.array_equal(it.cpu().numpy(), io) and np.array_equal(cv.cpu().numpy(), co)
        assert np.array_equal(x.cpu().numpy(), xo, equal_nan=True)
        assert len(np.unique(io)) > 3


def test_admm_frame_repack_is_bit_transparent(monkeypatch):
    # frames leave one by one (src/admm.py:65-66) while the non-converging ones run into the cap: live frames are re-gathered into
    # dense tiles (ldpc_repack.hpp).  Moving state must not change a bit: repack on == repack off == the C oracle.
    import torch
    from ldpc_decoders_amd._device import AdmmHandle

    monkeypatch.setenv("LDPC_ADMM_BACKEND", "stream")  # the repack belongs to the streaming kernels (the LDS-resident kernel has no tiles)
    rng = np.random.RandomState(9)
    for name, B, snr, max_iter, with_oracle in (("1200_3_6_rand_ldpc_1", 64 * 9 + 5, 2.3, 200, 96), ("1200_rho_x5_rand_ldpc_5", 64 * 6, 2.0, 120, 0),
                                                ("7_4_hamming", 64 * 40 + 3, 2.0, 100, 64 * 40 + 3)):
        code = _code(name)
        gamma = -2 * (-1 + rng.normal(0, np.sqrt(10 ** (-snr / 10)), (B, code.n))) / 10 ** (-snr / 10)
        gd = torch.from_numpy(gamma).cuda()
        outs = {}
        for mode, fill in (("0", None), ("1", "0.95"), ("1", None)):
            monkeypatch.setenv("LDPC_STREAM_REPACK", mode)
            if fill:
                monkeypatch.setenv("LDPC_STREAM_REPACK_FILL", fill)
            else:
                monkeypatch.delenv("LDPC_STREAM_REPACK_FILL", raising=False)
            h = AdmmHandle(code)
            x, it, cv = h.decode_device(gd, 3.0, 1e-5, max_iter)
            outs[(mode, fill)] = (x.cpu().numpy(), it.cpu().numpy(), cv.cpu().numpy(), h.last_repacks())
        ref = outs[("0", None)]
        assert ref[3] == 0 and outs[("1", "0.95")][3] >= 2
        for key in (("1", "0.95"), ("1", None)):
            assert np.array_equal(outs[key][0], ref[0], equal_nan=True) and np.array_equal(outs[key][1], ref[1]) and np.array_equal(outs[key][2], ref[2])
        assert len(np.unique(ref[1])) > 5
        if with_oracle:
            xo, io, co = A.admm_decode(graph_of(name), gamma[:with_oracle], 3.0, 1e-5, max_iter)
            assert np.array_equal(ref[1][:with_oracle], io) and np.array_equal(ref[0][:with_oracle], xo, equal_nan=True)


def test_admm_lds_resident_kernel_equals_streaming_kernels_and_oracle(monkeypatch):
    """The LDS-resident kernel (one workgroup per frame; codes whose checks all have six edges) against the streaming kernels on the same
    frames -- estimates, iteration counts, convergence flags bit for bit -- and a prefix against the C oracle: frames leaving at every
    iteration count, frames that run into the cap, max_iter = 1 and an uncapped run, batches that do not fill the chip and that do."""
    import torch
    from ldpc_decoders_amd._device import AdmmHandle

    rng = np.random.RandomState(21)
    for name, B, snr, max_iter, with_oracle in (("1200_3_6_rand_ldpc_1", 777, 2.2, 120, 64), ("512_3_6_rand_ldpc_2", 1500, 2.6, 80, 64),
                                                ("1200_3_6_rand_ldpc_2", 300, 3.0, 1, 32), ("1200_3_6_rand_ldpc_3", 130, 3.2, -1, 32)):
        code = _code(name)
        gamma = -2 * (-1 + rng.normal(0, np.sqrt(10 ** (-snr / 10)), (B, code.n))) / 10 ** (-snr / 10)
        gd = torch.from_numpy(gamma).cuda()
        got = {}
        for mode in ("lds", "stream"):
            if mode == "stream":
                monkeypatch.setenv("LDPC_ADMM_BACKEND", "stream")
            else:
                monkeypatch.delenv("LDPC_ADMM_BACKEND", raising=False)
            h = AdmmHandle(code)
            x, it, cv = h.decode_device(gd, 3.0, 1e-5, max_iter)
            got[mode] = (x.cpu().numpy(), it.cpu().numpy(), cv.cpu().numpy())
            assert h.last_backend() == mode
        assert np.array_equal(got["lds"][1], got["stream"][1]) and np.array_equal(got["lds"][2], got["stream"][2])
        assert np.array_equal(got["lds"][0], got["stream"][0], equal_nan=True)
        if max_iter > 1:
            assert len(np.unique(got["lds"][1])) > 5
        xo, io, co = A.admm_decode(graph_of(name), gamma[:with_oracle], 3.0, 1e-5, max_iter)
        assert np.array_equal(got["lds"][1][:with_oracle], io) and np.array_equal(got["lds"][2][:with_oracle], co)
        assert np.array_equal(got["lds"][0][:with_oracle], xo, equal_nan=True)


def _admm_main_cases():
    with open(os.path.join(GOLDEN, "main_counters_admm.json")) as fp:
        return json.load(fp)


@pytest.mark.parametrize("run", _admm_main_cases(), ids=lambda r: r["argline"].replace(" ", "_")[:60])
def test_exact_mode_reproduces_reference_admm_counters(run, tmp_path):
    from ldpc_decoders_amd import main

    argv = run["argline"].split() + ["--data_dir", str(tmp_path), "--console", "--exact", "--np-seed", str(run["seed"])]
    main.main(argv)
    with open(os.path.join(str(tmp_path), run["file_name"])) as fp:
        got = json.load(fp)
    want = run["result"]
    assert list(got) == list(want)
    for key in ("tot", "wec", "bec"):
        assert got[key] == want[key]
    for prm in want["dec"]:
        assert got["dec"][prm]["iter"] == want["dec"][prm]["iter"]  # the decoder's iteration histogram (src/admm.py:38-40)


def test_admm_device_monte_carlo_cli(tmp_path):
    # device-noise mode (channel kernel -> ADMM -> counters on the GPU) for all three channels; rates against the reference's
    # own seeded golden runs (tests/golden/main_counters_admm.json: bsc p=.1 WER 40/136, biawgn 2 dB 40/343, bec .3 40/193)
    from ldpc_decoders_amd import main

    for ch, prm, lo, hi, extra in (("bsc", 0.1, 0.2, 0.4, []), ("biawgn", 2.0, 0.08, 0.16, []), ("bec", 0.3, 0.14, 0.28, ["--allow-pseudo"])):
        res = main.main(("%s 7_4_hamming ADMM --codeword 0 --min-wec 2000 --max-iter 100 --params %g --batch 8192" % (ch, prm)).split()
                        + extra + ["--data_dir", str(tmp_path), "--console"])
        r = res[prm]
        assert r["wec"] >= 2000 and r["tot"] % 8192 == 0 and lo < r["wer"] < hi, (ch, r)
        assert sum(r["dec"]["iter"]) == r["tot"]  # every decoded frame is in the iteration histogram


def test_admm_edge_cases():
    # empty batch, one frame, max_iter = 1 (every frame leaves through the cap with the x of the first iteration), bad arguments
    import torch
    from ldpc_decoders_amd._device import AdmmHandle
    from ldpc_decoders_amd._lib import LdpcHipError

    code = _code("7_4_hamming")
    g = graph_of("7_4_hamming")
    h = AdmmHandle(code)
    x, it, cv = h.decode_device(torch.zeros((0, 7), dtype=torch.float64, device="cuda"), 3.0, 1e-5, 10)
    assert x.shape == (0, 7) and it.numel() == 0
    gamma = np.random.RandomState(2).normal(0, 3, (65, 7))
    for mi in (1, 2, 0):
        xo, io, co = A.admm_decode(g, gamma, 3.0, 1e-5, mi)
        x, it, cv = h.decode_device(torch.from_numpy(gamma).cuda(), 3.0, 1e-5, mi)
        assert np.array_equal(x.cpu().numpy(), xo) and np.array_equal(it.cpu().numpy(), io) and np.array_equal(cv.cpu().numpy(), co)
    with pytest.raises(LdpcHipError):
        h.decode_device(torch.from_numpy(gamma).cuda(), 0.0, 1e-5, 10)
    with pytest.raises(ValueError):
        h.decode_device(torch.from_numpy(gamma[:, :6].copy()).cuda(), 3.0, 1e-5, 10)
