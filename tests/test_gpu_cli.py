"""GPU tests of the experiment driver (ldpc_decoders_amd.main): --exact reproduces the reference's counters for a fixed
np.random seed; device-noise mode writes the same JSON schema and statistically consistent error rates."""
import json
import os

import numpy as np
import pytest

from helpers import CODES_DIR, GOLDEN, main_counter_cases

pytestmark = pytest.mark.gpu


def _is_llr_spa(run):
    return " SPA " in run["argline"] and not run["argline"].startswith("bec")


def _run_exact(run, tmp_path, monkeypatch):
    from ldpc_decoders_amd import codes, main

    monkeypatch.setenv(codes.file_codes_dir_string, CODES_DIR)
    argv = run["argline"].split() + ["--data_dir", str(tmp_path), "--console", "--exact", "--np-seed", str(run["seed"])]
    main.main(argv)
    with open(os.path.join(str(tmp_path), run["file_name"])) as fp:
        return json.load(fp)


@pytest.mark.parametrize("run", [r for r in main_counter_cases() if not _is_llr_spa(r)], ids=lambda r: r["argline"].replace(" ", "_")[:60])
def test_exact_mode_reproduces_reference_counters(run, tmp_path, monkeypatch):
    # reference: `np.random.seed(s); python src/main.py <argline>` (tests/golden/main_counters.json); min-sum, the erasure
    # decoder and every multi-parameter line (one np.random stream runs through all --params values)
    got = _run_exact(run, tmp_path, monkeypatch)
    want = run["result"]
    assert list(got) == list(want)
    for key in ("tot", "wec", "bec"):
        assert got[key] == want[key]
    for key in ("wer", "ber"):
        for prm in want[key]:
            assert got[key][prm] == pytest.approx(want[key][prm], rel=1e-12)


SPA_RUNS = [r for r in main_counter_cases() if _is_llr_spa(r)]


@pytest.mark.parametrize("run", SPA_RUNS, ids=lambda r: ("config1_" if "7_4_hamming" in r["argline"] else "") + r["argline"].replace(" ", "_")[:60])
def test_exact_mode_sum_product_lines_through_hip(run, tmp_path, monkeypatch, capsys):
    # The LLR sum-product arg-lines of tests/golden/main_counters.json -- BASELINE config 1's own anchor
    # `biawgn 7_4_hamming SPA --codeword 1 --min-wec 50 --max-iter 10 --params 2 4` -> 412/50/118 and 2025/50/133,
    # `biawgn 1200_3_6_rand_ldpc_1 SPA 1.5 dB` -> 20/5/331, `bsc 1200_3_6_rand_ldpc_1 SPA .07` -> 29/5/305 -- through
    # `ldpc_decoders_amd.main --exact` on the HIP path (fp64 sum-product, the reference formula verbatim, src/bpa.py:66-75).
    # Bar: tot, wec and bec EQUAL the reference's.  Device libm (tanh/log/exp/atanh) differs from numpy's by ulps, which could move
    # bits of a frame that does not converge upstream either; so the frames are also re-decoded one by one against the numpy
    # oracle on the same np.random stream, and the report says WHICH frames differ (none, measured) before the counters are compared.
    import bp_oracle as O
    from helpers import golden_edges
    from ldpc_decoders_amd import codes
    from ldpc_decoders_amd.models import models

    got = _run_exact(run, tmp_path, monkeypatch)
    want = run["result"]
    assert list(got) == list(want)
    a = run["argline"].split()
    channel, code_name = a[0], a[1]
    opt = {a[i]: a[i + 1] for i in range(3, len(a) - 1) if a[i].startswith("--") and a[i] != "--params"}
    cw, max_iter = int(opt["--codeword"]), int(opt["--max-iter"])
    g = golden_edges(code_name)
    code = codes.get_code(code_name)
    np.random.seed(run["seed"])
    report = []
    for prm in want["tot"]:
        tot = want["tot"][prm]
        x = np.full(g.n, cw, dtype=np.int64)
        send = {"biawgn": O.biawgn_send, "bsc": O.bsc_send}[channel]
        Y = np.stack([send(x, float(prm)) for _ in range(tot)])  # the reference's frames of this point, in its order
        Xo, Io = O.channel_decode(g, channel, "SPA", float(prm), Y, max_iter)
        dec = getattr(models[channel], "SPA")(float(prm), code, max_iter=max_iter, precision="f64")
        Xd, Id = dec.decode_batch(Y)
        Xd = np.asarray(Xd).astype(np.int64)
        differ = np.flatnonzero((Xd != Xo).any(axis=1))
        err_o, err_d = (Xo != x).sum(axis=1), (Xd != x).sum(axis=1)
        report.append("%s param %s: %d frames, device == oracle on %d (%.2f %%); differing frames %s, oracle iterations there %s; "
                      "counters device %d/%d/%d reference %d/%d/%d" % (run["argline"][:40], prm, tot, tot - len(differ), 100.0 * (tot - len(differ)) / tot,
                                                                         differ.tolist(), Io[differ].tolist(), got["tot"][prm], got["wec"][prm],
                                                                         got["bec"][prm], tot, want["wec"][prm], want["bec"][prm]))
        # the numpy oracle reproduces the reference's counters on these frames (pins the comparison itself)
        assert (int((err_o > 0).sum()), int(err_o.sum())) == (want["wec"][prm], want["bec"][prm])
        # a frame may differ only where upstream does not converge (it is a word error on both sides)
        assert (Io[differ] >= max_iter).all() and (err_o[differ] > 0).all() and (err_d[differ] > 0).all()
        assert len(differ) <= max(1, tot // 10)
    with capsys.disabled():
        print("\n" + "\n".join(report))
    # measured on MI355X (ROCm 7.2 device libm): every frame of every line identical to the numpy oracle, all counters equal
    for key in ("tot", "wec", "bec"):
        assert got[key] == want[key]
    if "7_4_hamming" in run["argline"]:  # BASELINE config 1, by name
        assert got["tot"] == {"2.0": 412, "4.0": 2025} and got["wec"] == {"2.0": 50, "4.0": 50} and got["bec"] == {"2.0": 118, "4.0": 133}


def test_device_mode_schema_and_rates(tmp_path, monkeypatch):
    from ldpc_decoders_amd import codes, main

    monkeypatch.setenv(codes.file_codes_dir_string, CODES_DIR)
    argv = "biawgn 1200_3_6_rand_ldpc_1 MSA --codeword 0 --min-wec 200 --max-iter 10 --params 2.0 2.5 --batch 8192".split()
    res = main.main(argv + ["--data_dir", str(tmp_path), "--console"])
    data = json.load(open(os.path.join(str(tmp_path), "biawgn-1200_3_6_rand_ldpc_1-MSA-0-200-10.json")))
    assert list(data)[:6] == ["channel", "code", "decoder", "codeword", "min_wec", "max_iter"]
    # published curve of this very code/decoder/max_iter (reference data/output/biawgn-1200_3_6_rand_ldpc_1-MSA-10-1.json):
    # BER 2.18e-2 @ 2.0 dB (140 frames), 2.37e-3 @ 2.5 dB (626 frames) -- same order of magnitude expected
    assert 0.012 < data["ber"]["2.0"] < 0.035 and 0.0012 < data["ber"]["2.5"] < 0.0045
    assert data["wec"]["2.0"] >= 200 and data["tot"]["2.0"] % 8192 == 0
    assert res[2.0]["wer"] == data["wer"]["2.0"]


def test_fp16_storage_mode_through_the_cli(tmp_path, monkeypatch):
    # `--precision f16` (an addition): the same run on the fp16-storage streaming kernels -- a tolerance mode; the rates must sit where
    # the fp32 run's do (same published curve as above)
    from ldpc_decoders_amd import codes, main

    monkeypatch.setenv(codes.file_codes_dir_string, CODES_DIR)
    argv = "biawgn 1200_3_6_rand_ldpc_1 MSA --codeword 0 --min-wec 200 --max-iter 10 --params 2.0 2.5 --batch 8192 --precision f16".split()
    main.main(argv + ["--data_dir", str(tmp_path), "--console"])
    data = json.load(open(os.path.join(str(tmp_path), "biawgn-1200_3_6_rand_ldpc_1-MSA-0-200-10.json")))
    assert 0.012 < data["ber"]["2.0"] < 0.035 and 0.0012 < data["ber"]["2.5"] < 0.0045
    assert data["wec"]["2.0"] >= 200 and data["tot"]["2.0"] % 8192 == 0


def test_curves_overlap_the_published_reference_results(tmp_path):
    # device Monte-Carlo against the reference's own published result files for the fixture codes (a subset here; the full
    # table is profiles/curves_vs_reference.md, tools/compare_curves.py): every curve agrees or is a documented deviation
    import subprocess
    import sys

    root = os.path.dirname(GOLDEN.rstrip("/")).rsplit("/tests", 1)[0]
    for sel in ("1200_3_6_rand_ldpc_1", "bsc-1200_3_6_ldpc-MSA-40", "7_4_hamming-ML", "bsc-1200_3_6_ldpc-SPA-40"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "compare_curves.py"), "--only", sel, "--min-wec", "600",
                            "--out", str(tmp_path / "c.md")], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "agrees" in open(str(tmp_path / "c.md")).read()


def test_max_frames_stops_a_low_error_point(tmp_path, monkeypatch):
    # --max-frames (an addition): a parameter whose word-error rate is too low to collect --min-wec errors stops at the cap --
    # the reference's tables go down to WER ~1e-9, where its own loop would run for years (src/main.py:37)
    from ldpc_decoders_amd import codes, main as M

    monkeypatch.setenv(codes.file_codes_dir_string, CODES_DIR)
    res = M.main(["bec", "1200_3_6_rand_ldpc_1", "SPA", "--codeword", "0", "--min-wec", "100000", "--max-iter", "10", "--params", "0.2",
                  "--batch", "8192", "--max-frames", "20000", "--data_dir", str(tmp_path), "--console"])
    r = res[0.2]
    assert r["tot"] == 3 * 8192 and r["wec"] < 100000  # whole rounds: the first total at or above the cap


def test_bench_lines_of_the_admm_and_ml_decoders():
    # bench.py --decoder ADMM | ML: a gradeable line each -- whole-step frames/s, the dominant kernel timed alone with
    # HIP events, its name (the LDS-resident ADMM kernel where the code is eligible), counters against published rates
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def line(flags):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + flags + ["--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        rows = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(rows) == 1
        return json.loads(rows[0])

    a = line(["--decoder", "ADMM", "--code", "512_3_6_rand_ldpc_2", "--param", "2.6", "--max-iter", "80", "--batch", "4096", "--steps", "2", "--warmup", "1",
              "--repeats", "1"])
    assert a["config"]["decoder"] == "ADMM" and a["config"]["backend"] == "admm-lds" and a["dtype"] == "f64" and a["frames_counted"] == 2 * 4096
    assert a["roofline"]["kernel"] == "k_admm_lds<6, 3, 2, 4, 1>" and a["roofline"]["avg_launch_ms"] > 0 and a["roofline"]["unit_of_work"] == "frame-iteration"
    assert 1 < a["mean_sweeps"] < 80 and 0 < a["wer"] < 1 and "baseline_configs" not in a
    m = line(["--decoder", "ML", "--code", "7_4_hamming", "--param", "2.0", "--precision", "f32", "--batch", "1048576", "--steps", "3", "--warmup", "1",
              "--repeats", "1", "--max-iter", "0"])
    assert m["config"]["decoder"] == "ML" and m["roofline"]["kernel"] == "k_ml<float, 0>" and m["frames_counted"] == 3 * 1048576
    assert abs(m["wer"] - 0.0910) < 0.004     # ML word-error rate of Hamming(7,4) at 2 dB (published curve: tests/golden/published_curves.json)
