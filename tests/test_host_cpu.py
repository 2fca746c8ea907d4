"""CPU-side tests: the C ABI library loads and exports every symbol of include/ldpc_hip.h, the code store mirrors the
reference loader, and the CLI / JSON surface keeps the reference's schema.  No GPU compute calls here."""
import json
import os
import re

import numpy as np
import pytest

from helpers import CODES_DIR, GOLDEN, golden_edges

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    with open(os.path.join(ROOT, "include", "ldpc_hip.h")) as fp:
        text = fp.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ldpc_[a-z0-9_]+)\s*\(", text)))


def test_c_abi_exports_every_declared_symbol():
    import ctypes

    from ldpc_decoders_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), "symbol %s declared in include/ldpc_hip.h but not exported" % name
    # the ctypes binding covers the same set
    assert set(_lib.SIGNATURES) == set(names)
    assert _lib.load().ldpc_abi_version() == 4


def test_argument_errors_are_reported_not_thrown():
    import ctypes

    from ldpc_decoders_amd import _lib

    lib = _lib.load()
    h = ctypes.c_void_p()
    chk = np.array([0, 0, 1], dtype=np.int32)
    var = np.array([1, 0, 1], dtype=np.int32)  # not row-major sorted
    rc = lib.ldpc_code_create(0, 2, 2, 3, chk.ctypes.data, var.ctypes.data, ctypes.byref(h))
    assert rc == -3 and b"row-major" in lib.ldpc_last_error()
    rc = lib.ldpc_code_create(0, 0, 2, 3, chk.ctypes.data, var.ctypes.data, ctypes.byref(h))
    assert rc == -1
    with pytest.raises(_lib.LdpcHipError):
        _lib.check(rc)
    # the round-5 entry points refuse null handles / buffers with LDPC_E_ARG before anything touches a device
    buf = (ctypes.c_uint64 * 8)()
    assert lib.ldpc_decode_bits(None, None, None, 4, 10, 0, buf, None, buf, None) == -1 and b"ldpc_decode_bits" in lib.ldpc_last_error()
    assert lib.ldpc_decode_host_bits(None, None, None, 4, 10, 0, buf, None, buf) == -1
    assert lib.ldpc_count_errors_bits(None, None, None, 0, None, 4, 7, 0, buf, None) == -1
    assert lib.ldpc_simulate_rounds(None, 0, 1.0, 0, 1, 0, 0, 64, 2, 64, 10, 0, 0, buf, None) == -1
    assert lib.ldpc_channel_list(0, 0, 1.0, 0, 1, 0, None, 4, 7, buf, None) == -1
    assert lib.ldpc_count_errors_list(None, 0, None, None, 4, 7, 0, buf, 4, 0, 0, 1, buf, None) == -1
    p = ctypes.c_void_p()
    assert lib.ldpc_decoder_grid_list(None, ctypes.byref(p), None, None) == -1 and lib.ldpc_decoder_grid_list_reset(None, None) == -1


@pytest.mark.parametrize("name", ["4_2_test", "6_2_3_ldpc", "7_4_hamming", "12_3_4_ldpc"])
def test_builtin_codes_match_reference(name):
    from ldpc_decoders_amd import codes

    c, g = codes.get_code(name), golden_edges(name)
    assert (c.m, c.n, c.E) == (g.m, g.n, g.E) and (c.edge_chk == g.chk).all() and (c.edge_var == g.var).all()
    cb = c.cb  # G H^T = 0 and the all-zero word first (src/codes.py:17-19)
    assert cb.shape == (2 ** c.gen_mtx.shape[0], c.n) and cb[0].sum() == 0 and c.syndrome(cb).sum() == 0
    assert (c.parity_mtx == g.to_dense()).all() and c.get_k() == c.n - c.m


@pytest.mark.parametrize("name", ["1200_3_6_rand_ldpc_1", "1200_rho_x5_rand_ldpc_5", "512_3_6_rand_ldpc_2", "margulis", "1200_3_6_ldpc"])
def test_file_loader_matches_reference(name, monkeypatch):
    from ldpc_decoders_amd import codes

    monkeypatch.setenv(codes.file_codes_dir_string, CODES_DIR)
    assert name in codes.get_code_names()
    c, g = codes.get_code(name), golden_edges(name)
    assert (c.m, c.n, c.E) == (g.m, g.n, g.E) and (c.edge_chk == g.chk).all() and (c.edge_var == g.var).all()


def test_code_names_resolve_without_environment(tmp_path, monkeypatch):
    """src/codes.py:68-90 resolves names against $FILE_CODES_DIR or ./data/codes; upstream is always run from its own checkout, so its
    27 data/codes files ship inside the package and are what a bare `python -m ldpc_decoders_amd.main ...` finds."""
    import subprocess
    import sys

    from ldpc_decoders_amd import codes, utils

    monkeypatch.delenv(codes.file_codes_dir_string, raising=False)
    monkeypatch.chdir(tmp_path)  # no ./data/codes here
    names = codes.get_code_names()
    assert len(names) == 4 + 27 and "1200_3_6_rand_ldpc_1" in names and "margulis" in names
    c, g = codes.get_code("1200_3_6_rand_ldpc_1"), golden_edges("1200_3_6_rand_ldpc_1")
    assert (c.edge_chk == g.chk).all() and (c.edge_var == g.var).all()
    # the argparse `choices` of the CLI (src/utils.py:24-27) accept the name from a clean environment
    parser = utils.setup_parser(codes.get_code_names(), ["biawgn", "bsc", "bec"], utils.decoder_names)
    args = parser.parse_args(["biawgn", "1200_3_6_rand_ldpc_1", "MSA", "--max-iter", "50", "--params", "2.0", "--console"])
    assert args.code == "1200_3_6_rand_ldpc_1"
    env = {k: v for k, v in os.environ.items() if k != codes.file_codes_dir_string}
    env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, "-c", "from ldpc_decoders_amd import codes; print(len(codes.get_code_names()), codes.get_code('margulis').n)"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == ["31", "2640"], r.stderr
    # ./data/codes of the working directory still shadows the packaged files, $FILE_CODES_DIR shadows both (upstream order)
    os.makedirs(tmp_path / "data" / "codes")
    (tmp_path / "data" / "codes" / "tiny.txt").write_text("1 2\n2 3\n")
    assert set(codes.get_code_names()) == {"4_2_test", "6_2_3_ldpc", "7_4_hamming", "12_3_4_ldpc", "tiny"}
    monkeypatch.setenv(codes.file_codes_dir_string, CODES_DIR)
    assert "margulis" in codes.get_code_names() and "tiny" not in codes.get_code_names()


def test_loader_edge_cases(tmp_path):
    from ldpc_decoders_amd import codes

    assert codes.parse_parity_text("1 2\n\n 2 3 3\n").E == 4  # blank lines skipped, duplicates collapse
    zero_based = codes.parse_parity_text("0 1\n1 2\n")  # var-1 quirk: variable 0 lands in the last column
    assert zero_based.n == 3 and sorted(zero_based.edge_var[zero_based.edge_chk == 0]) == [0, 2]
    with pytest.raises(Exception):
        codes.parse_parity_text("2 3\n3 4\n")
    np.random.seed(3)
    c = codes.rand_reg_ldpc(96, 3, 6)
    assert (c.row_degrees() == 6).all() and (c.col_degrees() == 3).all()
    path = codes.save_parity_mtx(c, "t_96_3_6", str(tmp_path))
    back = codes.load_parity_mtx(path)
    assert (back.edge_chk == c.edge_chk).all() and (back.edge_var == c.edge_var).all()


def test_cli_grammar_and_result_schema(tmp_path):
    from ldpc_decoders_amd import codes, utils
    from ldpc_decoders_amd.models import models

    p = utils.setup_parser(codes.get_code_names(), models.keys(), utils.decoder_names)
    # an arg-line as emitted by the reference's simulations.py (src/simulations.py:27-39)
    a = p.parse_args(("biawgn 7_4_hamming SPA --codeword=1 --min-wec=50 --max-iter=10 --params 2 4 --data_dir=%s --console" % tmp_path).split())
    assert a.params == [2.0, 4.0] and a.max_iter == 10 and a.min_wec == 50 and a.codeword == 1 and a.log_freq == 5.0
    with pytest.raises(SystemExit):
        p.parse_args("awgn 7_4_hamming SPA".split())
    ids = [("channel", "biawgn"), ("code", "7_4_hamming"), ("decoder", "SPA"), ("codeword", 1), ("min_wec", 50), ("max_iter", 10)]
    s = utils.Saver(str(tmp_path), ids)
    assert os.path.basename(s.file_path) == "biawgn-7_4_hamming-SPA-1-50-10.json"
    s.add(2.0, {"tot": 412, "wec": 50, "wer": 50 / 412, "bec": 118, "ber": 118 / (412 * 7)})
    s.add(4.0, {"tot": 2025, "wec": 50, "wer": 50 / 2025, "bec": 133, "ber": 133 / (2025 * 7)})
    data = json.load(open(s.file_path))
    assert list(data)[:6] == [k for k, _ in ids] and data["tot"] == {"2.0": 412, "4.0": 2025}
    assert set(data) == {"channel", "code", "decoder", "codeword", "min_wec", "max_iter", "tot", "wec", "wer", "bec", "ber"}


def test_registry_surface():
    from ldpc_decoders_amd.models import decoder_names, models

    assert set(models) == {"bsc", "bec", "biawgn"} and decoder_names == ["ML", "SPA", "MSA", "LP", "ADMM", "ADMMA"]
    for ch, mod in models.items():
        assert hasattr(mod, "Channel")
        for name in ("SPA", "MSA"):
            assert getattr(mod, name).id_keys == ["max_iter"]
        with pytest.raises(NotImplementedError):
            mod.LP(0.1, None, max_iter=1)
    np.random.seed(0)
    x = np.zeros(8, dtype=np.int64)
    assert set(np.unique(models["bec"].Channel(0.5).send(x))) <= {0, 2}
    assert set(np.unique(models["bsc"].Channel(0.5).send(x))) <= {0, 1}
    assert models["biawgn"].Channel(2.0).send(x).dtype == np.float64


@pytest.mark.parametrize("case", ["HMG", "MAR", "REG_BAD", "REG_ENS", "IREG_ENS"])
def test_simulation_case_tables_match_reference(case):
    # arg-lines printed by the reference's simulations.py (captured in tests/golden/simulations_lines.json)
    from ldpc_decoders_amd import simulations

    with open(os.path.join(GOLDEN, "simulations_lines.json")) as fp:
        want = json.load(fp)[case]
    extra = ["--data_dir=/tmp/x", "--console"]
    assert simulations.lines(case, extra, all_decoders=True) == want
    built = simulations.lines(case, extra)  # default: every decoder this build has (only LP lines are dropped)
    assert built == [ln for ln in want if ln.split()[2] != "LP"] and len(built) > 0
    if case in ("HMG", "MAR"):
        assert any(ln.split()[2] == "ADMM" for ln in built) and (case == "MAR" or any(ln.split()[2] == "ML" for ln in built))


def test_code_generator_cli(tmp_path):
    # python -m ldpc_decoders_amd.codes <count> <n> <l> <r>  (reference: python src/codes.py ..., src/codes.py:139-174)
    from ldpc_decoders_amd import codes

    np.random.seed(4)
    paths = codes.gen_rand_ldpc(codes.setup_parser().parse_args(["2", "96", "3", "6", "--dir", str(tmp_path)]))
    assert [os.path.basename(p) for p in paths] == ["96_3_6_rand_ldpc_1.txt", "96_3_6_rand_ldpc_2.txt"]
    c = codes.load_parity_mtx(paths[1])
    assert (c.m, c.n) == (48, 96) and set(c.col_degrees()) == {3} and set(c.row_degrees()) == {6}
    paths = codes.gen_rand_ldpc(codes.setup_parser().parse_args(["1", "1200", "--irregular", "--dir", str(tmp_path)]))
    c = codes.load_parity_mtx(paths[0])
    assert c.n == 1200 and c.col_degrees().max() == 8 and set(np.unique(c.row_degrees())) <= {2, 4, 6}


def test_product_fails_loudly_without_the_hip_library_or_a_gpu(tmp_path):
    # no CPU fallback anywhere: a missing libldpc_hip.so raises at load, and on a host without a GPU constructing a decoder
    # raises from the library (hipMalloc / hipSetDevice fails) instead of computing something on the CPU
    import subprocess
    import sys

    code = ("import os, sys; sys.path.insert(0, %r); os.environ['LDPC_LIB_PATH'] = %r\n"
            "from ldpc_decoders_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.LdpcHipError as e:\n    print('RAISED', 'no CPU fallback' in str(e))\n"
            % (ROOT, str(tmp_path / "missing.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr
    import torch

    if not torch.cuda.is_available():
        from ldpc_decoders_amd import _lib, bpa, codes

        with pytest.raises(_lib.LdpcHipError):
            bpa.MSA(codes.get_code("7_4_hamming"), max_iter=5)


def _check_bench_line(d, want_cpu=True):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["data"] == "synthetic"
    r = d["roofline"]
    assert r["bound"] in ("hbm", "lds", "valu") and r["unit"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert 0 < r["frac"] <= 1.0  # a fraction OF the binding resource (HBM for the streaming kernels, LDS pipe / VALU on chip), never the 8(d) HBM model of an on-chip kernel
    if r["bound"] != "hbm":
        assert r["binding_unit"] in ("lds_array", "lds_store_path", "valu") and r["hbm_model"]["flag"]
    if want_cpu:
        c = d["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["physical_cores"]


def test_committed_bench_lines_keep_the_contract():
    # every BASELINE configuration has a committed bench line of the last measured round carrying every field of the bench contract,
    # a roofline object AND a CPU baseline measured in the same run
    import glob

    tags = sorted({os.path.basename(f).split("_")[0] for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json"))})
    assert tags
    tag = tags[-1]
    need = ["bench", "bench_f32", "bench_config3_spa_bsc", "bench_config3_bec", "bench_config4", "bench_config4_f64_stream", "bench_config5",
            "bench_driver_command"]
    lines = {}
    for name in need:
        path = os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name))
        assert os.path.exists(path), path
        with open(path) as fp:
            lines[name] = json.load(fp)
        _check_bench_line(lines[name])
    head = lines["bench"]
    assert head["dtype"] == "f64" and head["value"] > 1e7  # the reference's arithmetic, >= 1e7 frames/s at 50 sweeps on one MI355X
    assert head["roofline"]["traffic"] and head["config"]["batch_per_gpu"] == 65536 and head["config"]["decoder"] == "MSA"
    spa, bec = lines["bench_config3_spa_bsc"], lines["bench_config3_bec"]
    assert spa["config"]["decoder"] == "SPA" and spa["config"]["channel"] == "bsc" and spa["config"]["batch_per_gpu"] == 65536 and spa["dtype"] == "f32"
    assert bec["config"]["decoder"] == "BEC" and bec["config"]["channel"] == "bec" and bec["config"]["batch_per_gpu"] == 65536
    assert bec["config"]["steps_per_launch"] in (8, 32) and bec["value"] > 4.0e8   # >= 420 M frames/s at the BASELINE batch
    assert spa["value"] > 2.7e7
    c4, c5 = lines["bench_config4"], lines["bench_config5"]
    assert c4["config"]["n"] == 10000 and c4["config"]["batch_per_gpu"] == 131072 and c4["config"]["backend"] == "fused"
    assert c5["config"]["n"] == 64800 and c5["config"]["batch_per_gpu"] == 32768 and c5["config"]["backend"] == "stream" and c5["roofline"]["bound"] == "hbm"
    # round 6: the line of the DRIVER's command carries configs 3-5 itself (`baseline_configs`), each entry with a roofline and a CPU baseline;
    # counters of every LDS-resident line were collected on the very kernel code that was timed; the SURVEY 8(f) decoders have lines of their own
    if tag >= "r06":
        drv = lines["bench_driver_command"]
        block = drv["baseline_configs"]
        assert list(block) == ["config3_spa_bsc", "config3_bec", "config4", "config5"]
        for name, e in block.items():
            assert "error" not in e and e["frames_per_s"] > 0 and e["n_gpus"] == 1, name
            assert e["roofline"]["frac"] and e["roofline"]["frac"] > 0.5 and e["roofline"].get("counters_stale") in (False, None), name
            assert e["cpu_baseline"]["kind"] == "port" and e["cpu_baseline"]["value"] > 0, name
        assert block["config5"]["frames_per_s"] > 1.06e5 and block["config3_bec"]["frames_per_s"] > 4.0e8
        for name in ("bench", "bench_f32", "bench_config3_spa_bsc", "bench_config3_bec", "bench_config4", "bench_driver_command"):
            assert lines[name]["roofline"]["counters_stale"] is False, name
        # >= 110 k, side kernels <= 16 ms.  The pool's boxes differ by +-2.3 % on this HBM-bound line (107.6 / 109.9 / 111.2 / 112.6 k
        # with the same kernels; round 5's 104.5 k was one box): the floor here is the slowest box seen, the second line another box's
        assert c5["value"] > 1.06e5 and c5["side_kernels_ms_per_step"] < 16.0
        other = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_config5_second_box.json" % tag)))
        assert other["config"] == c5["config"] and other["value"] > 1.10e5
        for name, alg in (("bench_admm", "ADMM"), ("bench_ml", "ML")):
            with open(os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name))) as fp:
                d = json.load(fp)
            _check_bench_line(d)
            assert d["config"]["decoder"] == alg and d["roofline"]["bound"] == "valu" and d["roofline"]["frac"] and d["roofline"]["counters_stale"] is False
            assert d["cpu_baseline"]["value"] > 0 and d["roofline"]["avg_launch_ms"] > 0
    # the strong-scaling harness lines (BASELINE's whole-node batches on the GPUs the box had)
    for name, total in (("bench_config4_total_batch", 1 << 20), ("bench_config5_total_batch", 1 << 18)):
        path = os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name))
        if os.path.exists(path):
            with open(path) as fp:
                d = json.load(fp)
            assert d["scaling"] == "strong" and d["config"]["total_batch"] == total and d["frames_counted"] == d["steps"] * total


@pytest.mark.parametrize("n", [8, 7])  # odd n: numpy's legacy normal() caches the second deviate of a pair across frames
def test_exact_mode_leaves_the_random_stream_where_the_reference_does(n):
    # src/main.py:37-40 draws one frame per trip; a chunked run that stops inside a chunk must rewind to the same stream position,
    # otherwise every later --params value sees different noise (the committed multi-parameter n=1200 goldens pin the GPU side)
    from ldpc_decoders_amd import biawgn
    from ldpc_decoders_amd.montecarlo import run_point_exact

    class Slicer:  # host-only stand-in for a decoder: hard decision on the received values
        def decode_batch(self, y):
            return (np.asarray(y) > 0).astype(np.uint8), np.ones(len(y), dtype=np.int32)

    x = np.zeros(n, dtype=np.int64)
    outs = []
    for chunk in (1, 5, 32):
        np.random.seed(4321)
        pts = [run_point_exact(biawgn.Channel(snr), Slicer(), x, 9, chunk=chunk) for snr in (0.0, 1.0, 2.0)]
        outs.append((pts, np.random.normal()))
    assert outs[0] == outs[1] == outs[2]
    assert outs[0][0][0]["tot"] % 5 != 0 or outs[0][0][1]["tot"] % 5 != 0  # the stop really fell inside a chunk


def test_irregular_ensembles_match_the_reference_node_counts():
    # tests/golden/irregular_ensembles.json: lambda(x) of ldpc.solve_dist and int(L_i n) of gen_L_R for rho = x^4, x^5, x^6
    # (captured from the reference by oracle/make_goldens_gen.py): 742/211/107/58/80 for the rho = x^5, n = 1200 ensemble
    from ldpc_decoders_amd import codes

    with open(os.path.join(GOLDEN, "irregular_ensembles.json")) as fp:
        want = json.load(fp)
    for rho in (4, 5, 6):
        lam = codes.LAMBDA_HALF_RATE[rho]
        ref = {int(k): v for k, v in want[str(rho)]["lambda_edge"].items()}
        assert sorted(lam) == sorted(ref) and all(abs(lam[d] - ref[d]) < 1e-4 for d in lam)
        for n in (1200, 10000):
            counts, extra = codes.irregular_degree_counts(n, lam, rho + 1)
            base = dict(counts)
            for d in extra:
                base[d] -= 1
            ref_counts = {int(k): v for k, v in want[str(rho)]["node_counts"][str(n)].items()}
            if (rho, n) == (5, 10000):  # the published 4-digit lambda of rho = x^5 moves a handful of nodes at n = 10 000
                assert sum(abs(base[d] - ref_counts[d]) for d in base) <= 6
            else:
                assert base == ref_counts
            assert sum(counts.values()) == n and sum(d * c for d, c in counts.items()) % (rho + 1) == 0
    assert {int(k): v for k, v in want["5"]["node_counts"]["1200"].items()} == {2: 742, 3: 211, 4: 107, 7: 58, 8: 80}
    code = codes.rand_irregular_ldpc(1200, codes.LAMBDA_HALF_RATE[6], 7, np.random.RandomState(15))
    assert code.n == 1200 and code.row_degrees().max() == 7 and code.col_degrees().max() <= 16


def test_host_only_layout_planning_and_plan_store(tmp_path):
    # ldpc_plan_layout needs no GPU: shape choice + annealed LDS placement, stored as <key>.plan; a second call with the store
    # directory on the search path would load it.  The plan of the headline code must beat the trivial placement by far.
    import ctypes

    from ldpc_decoders_amd import _lib, codes

    lib = _lib.load()
    code = codes.load_parity_mtx(os.path.join(CODES_DIR, "1200_3_6_rand_ldpc_1.txt"))
    chk = np.ascontiguousarray(code.edge_chk, dtype=np.int32)
    var = np.ascontiguousarray(code.edge_var, dtype=np.int32)
    info = (ctypes.c_double * 4)()
    # fp32 (all algorithms share it), fp64 min-sum (four waves per frame), fp64 sum-product (fixed edge order)
    for alg, dtype, waves in ((0, 0, 2), (0, 1, 4), (1, 1, 2)):
        out = tmp_path / ("plans_%d_%d" % (alg, dtype))
        out.mkdir()
        _lib.check(lib.ldpc_plan_layout(code.m, code.n, code.E, chk.ctypes.data, var.ctypes.data, alg, dtype, 400000, str(out).encode(), info))
        assert info[0] == waves and info[1] == 240 and 500 < info[2] < 600  # waves per frame, conflict-free gather cycles, trivial placement
        assert 0 < info[3] < 0.75 * info[2]
        files = os.listdir(out)
        assert len(files) == 1 and files[0].endswith(".plan") and os.path.getsize(out / files[0]) > 4 * (code.m + code.n + 2 * code.E)
    # an irregular code of the reference's rate-1/2 ensemble goes onto the shape with PAIR rounds (per wave two rounds of eight gathers, two
    # of three, six of two: 2 x (60 + 2 x 34) conflict-free cycles per sweep instead of 2 x (60 + 2 x 40)); a variable sits in a round at
    # least as wide as its degree
    irr = codes.load_parity_mtx(os.path.join(CODES_DIR, "1200_rho_x5_rand_ldpc_3.txt"))
    _lib.check(lib.ldpc_plan_layout(irr.m, irr.n, irr.E, np.ascontiguousarray(irr.edge_chk, dtype=np.int32).ctypes.data,
                                    np.ascontiguousarray(irr.edge_var, dtype=np.int32).ctypes.data, 0, 0, 200000, None, info))
    assert info[0] == 2 and info[1] == 256 and 0 < info[3] < info[2]
    # a graph no fused shape exists for: reported as such, nothing written
    big = codes.rand_reg_ldpc(20000, 3, 6, np.random.RandomState(1))
    _lib.check(lib.ldpc_plan_layout(big.m, big.n, big.E, np.ascontiguousarray(big.edge_chk, dtype=np.int32).ctypes.data,
                                    np.ascontiguousarray(big.edge_var, dtype=np.int32).ctypes.data, 0, 0, 1000, None, info))
    assert info[0] == 0
    # a malformed edge list is refused with the graph error of ldpc_code_create
    rc = lib.ldpc_plan_layout(2, 2, 3, np.array([0, 1, 0], dtype=np.int32).ctypes.data, np.array([0, 1, 1], dtype=np.int32).ctypes.data, 0, 0, 1000, None, info)
    assert rc == -3 and b"row-major" in lib.ldpc_last_error()


def test_nothing_throws_across_the_c_boundary():
    """include/ldpc_hip.h: "nothing throws across the boundary".  Every extern "C" entry runs inside a guard that turns a C++ exception
    of the host containers (edge lists, layout planner, plan files) into a negative code + ldpc_last_error(); absurd sizes come back as
    an error code, never as an abort of the calling Python process."""
    import ctypes

    from ldpc_decoders_amd import _lib

    lib = _lib.load()
    info = (ctypes.c_double * 4)()
    chk = np.zeros(4, dtype=np.int32)
    # E = 2^40 edges claimed with 4 readable: refused before any container is sized from it
    rc = lib.ldpc_plan_layout(2, 2, 1 << 40, chk.ctypes.data, chk.ctypes.data, 0, 0, 1000, None, info)
    assert rc < 0 and b"bad graph arguments" in lib.ldpc_last_error()
    for m, n, E in ((-1, 4, 4), (4, 0, 4), (4, 4, -7)):
        assert lib.ldpc_plan_layout(m, n, E, chk.ctypes.data, chk.ctypes.data, 0, 0, 1000, None, info) < 0
    # the guard is on every int-returning entry point of the shared object
    src = open(os.path.join(ROOT, "ldpc_decoders_amd", "csrc", "ldpc_api.hip")).read()
    body = src[src.index('extern "C" {'):]
    entries = [ln for ln in body.splitlines() if ln.startswith("int ldpc_") and "ldpc_abi_version" not in ln]
    assert len(entries) >= 30 and body.count("return guarded(") == len(entries)


# The Monte-Carlo (SIM) kernels behind `ldpc_simulate` for BASELINE configs 2-4 (+ the sum-product / irregular siblings main.py runs
# by default): template arguments <ALG, DC, DV, CRW, VRW, NW, SIM, VRX, DVX> of csrc/ldpc_fused_kernels.hpp
SIM_KERNELS_WITHOUT_SPILLS = [
    "k_fused_bp<0, 6, 3, 5, 10, 2, true, 0, 3>",    # config 2, fp32 min-sum, n = 1200 (3,6): the kernel `bench.py --precision f32` times
    "k_fused_f64<0, 6, 3, 3, 5, 4, true, 0, 3>",    # config 2, fp64 min-sum: the kernel `bench.py` times by default (four waves per frame)
    "k_fused_f64<0, 6, 3, 5, 10, 2, true, 0, 3>",   # its two-wave sibling (LDPC_FUSED_NW=2; the shape of fp64 sum-product)
    "k_fused_f64<1, 6, 3, 5, 10, 2, true, 0, 3>",   # config 3, fp64 sum-product (the reference's chain; branch-free rule, ldpc_cn.hpp)
    "k_fused_bp<0, 6, 3, 5, 10, 2, true, 2, 8>",    # irregular n = 1200 ensembles (1200_rho_x5_*), min-sum
    "k_fused_bp<0, 6, 3, 4, 8, 1, true, 0, 3>",     # n <= 512
    "k_fused_bp<0, 6, 3, 6, 11, 4, true, 0, 3>",    # Margulis n = 2640
    "k_fused_becs_mc<6, 3, 3, 5, 4, 0, 3>",        # config 3, erasure decoder, bit-sliced (csrc/ldpc_bec_kernels.hpp: no ALG parameter)
]


# One Monte-Carlo kernel is deliberately NOT in that list: a spill-free form exists (round 3 shipped it) and is measurably
# slower than the compiler's own allocation, which unpacks gather addresses once per frame, spills some and reloads them per sweep on the
# otherwise idle vector-memory pipe (same-box, same plans, profiles/r03C_spill_or_unpack.txt).  Bounded here so that a regression shows.
SIM_KERNELS_WITH_A_SPILL_BUDGET = {
    "k_fused_bp<0, 6, 3, 5, 10, 16, true, 3, 8>": 24,  # the two-width 16-wave shape (fallback since round 6): 19 spilled, 17.86 ms against 18.70 (spill-free) per 32 768 frames
    # round 6, shapes with pair rounds (VRX argument 98 = 2 wide + 16 x 6 pair rounds): config 4's kernel (6 spilled; every spill-free setting
    # measured 3-10 % slower), its guarded sibling (12 + 12 packed table words, one-instruction unpack), and the two-wave kernels of the
    # reference's irregular n = 1200 codes (4 + 4 / 8 + 8 packed words: HISTORY.md round 6)
    "k_fused_bp<0, 6, 3, 5, 10, 16, true, 98, 8>": 8,
    "k_fused_bp_grid<6, 3, 5, 10, 16, true, 98, 8>": 12,
    "k_fused_bp<0, 6, 3, 5, 10, 2, true, 98, 8>": 4,
    "k_fused_bp<1, 6, 3, 5, 10, 2, true, 98, 8>": 16,
    # config 3, fp32 sum-product (BSC / BI-AWGN), round 5: with the pair-tree rule the kernel is fastest with 6 + 8 of its gather-table words
    # kept packed and 6 registers spilled (2.21 ms per 65 536 frames at p = 0.07); every spill-free setting (15 + 15: 2.34 ms) is 5.7 %
    # slower, 0 + 0 / 4 + 4 / 8 + 8 / 10 + 10 lie in between -- same box, round 5 (HISTORY.md)
    "k_fused_bp<1, 6, 3, 5, 10, 2, true, 0, 3>": 8,
}


def test_simulate_kernels_do_not_spill():
    """Code-object metadata of the BUILT library (llvm-readelf --notes through tools/kernel_resources.py, no GPU needed): the named
    Monte-Carlo kernels keep everything in registers -- no spilled VGPR, no scratch segment.  (Round 2 shipped them with 24-37 spilled
    registers: 86 MB of scratch traffic per launch of a kernel whose only output is 55 counters.)"""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources

    ks = kernel_resources.kernels_of()
    assert len(ks) > 100, "code objects of libldpc_hip.so not found"
    # (not in the list: the check-degree 4/5/7/8 shapes)
    by_name = {k.split("(")[0]: v for k, v in ks.items()}
    for name in SIM_KERNELS_WITHOUT_SPILLS:
        assert name in by_name, "kernel %s not in the library" % name
        r = by_name[name]
        assert r["spill"] == 0 and r["scratch"] == 0, "%s: %d spilled VGPRs, %d B of scratch per lane" % (name, r["spill"], r["scratch"])
    for name, budget in SIM_KERNELS_WITH_A_SPILL_BUDGET.items():
        assert name in by_name, "kernel %s not in the library" % name
        assert by_name[name]["spill"] <= budget and by_name[name]["vgpr"] <= 128, "%s: %s" % (name, by_name[name])
    for name in SIM_KERNELS_WITHOUT_SPILLS:
        r = by_name[name]
        if (name.startswith("k_fused_bp") and (", 2, true" in name or ", 16, true" in name)) or name.startswith("k_fused_f64<0, 6, 3, 3, 5, 4") or name.startswith("k_fused_becs_mc<6, 3, 3, 5, 4"):
            assert r["vgpr"] <= 128  # four waves per SIMD: the occupancy the fp32 multi-wave shapes and the four-wave fp64 shape are built for


def test_compiler_never_touches_m0_in_the_fused_kernels():
    """The fused kernels set M0 with inline asm for their ds_write_addtid_b32 row stores (csrc/ldpc_fused_kernels.hpp: lds_set_m0 /
    lds_st_tid); the compiler does not know.  Sound only while the compiler has no M0 use of its own in those kernels -- checked on the
    disassembly of the BUILT library: every M0 reference is one of ours, and nothing that consumes M0 implicitly (movrel, sendmsg,
    LDS-DMA, GWS) was generated.  A compiler upgrade that breaks the assumption fails here instead of corrupting LDS rows."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources

    stores, foreign, consumers = kernel_resources.m0_report()
    assert stores > 1000, "ds_write_addtid_b32 stores not found in the library"
    assert not foreign, "M0 used outside lds_set_m0: %s" % foreign[:5]
    assert not consumers, "instructions that read M0 implicitly: %s" % consumers[:5]


def test_fused_kernels_reach_their_hand_off_words_as_lds():
    """Verdicts, frame numbers and error counts travel between the waves of a frame through LDS words (lds_word(),
    csrc/ldpc_fused_kernels.hpp).  A generic pointer there compiles to flat_store / flat_load sc0 sc1 in the sweep loop's serial tail."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources

    flat = kernel_resources.flat_report()
    assert not flat, "flat accesses in the fused kernels: %s" % flat[:5]


def test_bench_workload_selectors_and_cpu_baseline_dispatch():
    # bench.py --decoder / --channel / --param: the reference's selectors (src/main.py:11-12, src/models.py:3); the `bec` selector pairs
    # with the ternary erasure decoder whatever SPA / MSA says; the CPU baseline leg follows the same selectors (C port; scipy leg for
    # the LLR decoders) -- exercised here on the (7,4) Hamming code with a fraction of a second of CPU work
    import sys

    sys.path.insert(0, ROOT)
    import bench

    assert bench.resolve_workload("MSA", "biawgn", None, 1.0) == ("MSA", "biawgn", 1.0)
    assert bench.resolve_workload("SPA", "bsc", None, 1.0) == ("SPA", "bsc", 0.07)
    assert bench.resolve_workload("SPA", "bec", None, 1.0) == ("BEC", "bec", 0.40)
    assert bench.resolve_workload("MSA", "bec", 0.35, 1.0) == ("BEC", "bec", 0.35)
    assert bench.resolve_workload("BEC", "biawgn", None, 1.0)[:2] == ("BEC", "bec")
    code = bench.load_code("7_4_hamming")
    assert bench.bytes_per_frame_sweep(code, "MSA", "f32") == 4 * (4 * code.E + code.n)      # s(4E + n)
    assert bench.bytes_per_frame_sweep(code, "SPA", "f64") == 8 * (4 * code.E + code.n)
    assert bench.bytes_per_frame_sweep(code, "MSA", "f16") == 8 * code.E + 4 * code.n
    assert bench.bytes_per_frame_sweep(code, "BEC", "f32") == (4 * code.E + code.m + 3 * code.n) / 4.0
    for alg, channel, param in (("SPA", "bsc", 0.07), ("BEC", "bec", 0.40), ("MSA", "biawgn", 2.0)):
        c = bench.cpu_baseline(code, alg, channel, param, 10, "f32", 0.05)
        assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["physical_cores"] and channel in c["sample"]
        if alg == "BEC":
            assert "skipped" in c["scipy"]
        else:
            assert c["scipy"].get("value", 0) > 0, c["scipy"]
    args = bench.parse_args(["--decoder", "SPA", "--channel", "bec", "--total-batch", "1000", "--gpus", "2"])
    assert args.decoder == "SPA" and args.channel == "bec" and args.total_batch == 1000 and args.param is None and args.points is None


def test_bench_selectors_of_the_admm_and_ml_decoders():
    # bench.py --decoder ADMM | ML: selectors, byte models, CPU baselines (oracle/admm_oracle.c, oracle/ml_oracle.py) on
    # the (7,4) Hamming code with a fraction of a second of CPU work
    import sys

    sys.path.insert(0, ROOT)
    import bench

    assert bench.resolve_workload("ADMM", "biawgn", 2.2, 1.0) == ("ADMM", "biawgn", 2.2)
    assert bench.resolve_workload("ML", "bsc", None, 1.0) == ("ML", "bsc", 0.07)
    code = bench.load_code("7_4_hamming")
    assert bench.bytes_per_frame_sweep(code, "ADMM", "f64") == 8 * (9 * code.E + 2 * code.n)
    assert bench.bytes_per_frame_sweep(code, "ML", "f32") == 4 * code.n
    a = bench.cpu_baseline(code, "ADMM", "biawgn", 2.0, 50, "f64", 0.1)
    assert a["kind"] == "port" and a["value"] > 0 and a["cores"] >= 1 and "admm_oracle.c" in a["sample"] and "skipped" in a["scipy"]
    m = bench.cpu_baseline(code, "ML", "biawgn", 2.0, 0, "f32", 0.2)
    assert m["kind"] == "port" and m["value"] > 0 and m["cores"] == 1 and "16 codewords" in m["sample"]
    args = bench.parse_args(["--decoder", "ADMM", "--param", "2.2", "--max-iter", "300"])
    assert args.decoder == "ADMM" and not bench.is_default_workload(args)


def test_counter_entries_are_checked_against_the_kernel_code_of_the_built_library():
    # profiles/roofline_counters.json keys an LDS-resident kernel by NAME; a kernel body can change under an unchanged name.  Every entry
    # therefore carries a hash of the machine code it was measured on (tools/kernel_resources.py: ELF symbol bytes + kernel descriptor,
    # no GPU needed) and bench.py refuses to price a timing with counters of another body: counters_stale, frac null.
    import sys

    sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
    import bench
    import kernel_resources

    hashes = kernel_resources.kernel_code_hashes()
    headline = "k_fused_f64<0, 6, 3, 3, 5, 4, true, 0, 3>"
    assert len(hashes) > 200 and headline in hashes and "k_cn<float, 0, 6, 6, 2, false>" in hashes and "k_admm_lds<6, 3, 3, 8, 2>" in hashes
    assert all(len(h) == 16 for h in hashes.values())
    assert hashes[headline] != hashes["k_fused_f64<0, 6, 3, 3, 5, 4, false, 0, 3>"]      # the decode and the simulate variant differ
    assert kernel_resources.kernel_code_hashes() == hashes                               # a pure function of the file
    entry = {"lds_idx_active_per_frame_sweep": 600.0, "bank_conflict_per_frame_sweep": 30.0, "valu_active_cycles_per_frame_sweep": 1500.0,
             "insts_valu_per_frame_sweep": 700.0, "insts_lds_per_frame_sweep": 220.0, "lds_path_cycles_per_frame_sweep": 780.0}
    fresh = bench.fused_roofline(headline, 6.0e8, 256, {headline: dict(entry, kernel_code_sha=hashes[headline], head="abc")})
    assert fresh["counters_stale"] is False and fresh["frac"] is not None and fresh["binding_unit"] == "lds_store_path"
    stale = bench.fused_roofline(headline, 6.0e8, 256, {headline: dict(entry, kernel_code_sha="0123456789abcdef")})
    assert stale["counters_stale"] is True and stale["frac"] is None and stale["achieved"] is None and "STALE" in stale["counters_check"]
    unverified = bench.fused_roofline(headline, 6.0e8, 256, {headline: entry})
    assert unverified["counters_stale"] is None and unverified["frac"] is not None and "unverified" in unverified["counters_check"]
