"""N>1 path on CPU: world_size-2 gloo process group, frames sharded by global frame index, one all-reduce of counters.

The GPU decode is replaced by the CPU oracle here (this is a test of the sharding + collective plumbing of
ldpc_decoders_amd.dist / montecarlo, which is identical on RCCL); counters must not depend on the number of ranks.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path[:0] = [%(root)r, %(root)r + "/oracle", %(root)r + "/tests"]
import numpy as np
import bp_oracle as O, c_oracle as C
from helpers import golden_edges
from ldpc_decoders_amd import dist

comm = dist.init_from_env(prefer_gpu=False)
g = golden_edges("7_4_hamming")
seed, stream, snr, total, frame0 = 77, 2, 2.0, 1001, 5
start, cnt = comm.shard(frame0, total)
var = O.biawgn_noise_var(snr)
pri = np.stack([-2 * (-1 + np.sqrt(var) * O.device_biawgn_noise(seed, stream, f, g.n)) / var for f in range(start, start + cnt)]) if cnt else np.zeros((0, g.n))
xh, it = C.bp_decode(g, "MSA", None, pri, 10) if cnt else (np.zeros((0, g.n), np.uint8), np.zeros(0, np.int32))
err = (xh != 0).sum(axis=1)
local = np.array([cnt, (err > 0).sum(), err.sum(), it.sum()], dtype=np.int64)
red = comm.all_reduce_sum(local.copy())
mx = comm.max_float(float(comm.rank))
comm.barrier()
if comm.is_root:
    json.dump({"world": comm.world, "counters": [int(v) for v in red], "max_rank": mx, "shard0": [int(start), int(cnt)]}, open(sys.argv[1], "w"))
dist.finalize()
'''


def _run(world, out_path, port):
    code = WORKER % {"root": ROOT}
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", code, out_path], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    with open(out_path) as fp:
        return json.load(fp)


@pytest.mark.timeout(600)
def test_counters_independent_of_world_size(tmp_path):
    one = _run(1, str(tmp_path / "w1.json"), 29611)
    two = _run(2, str(tmp_path / "w2.json"), 29612)
    assert one["world"] == 1 and two["world"] == 2
    assert one["counters"] == two["counters"]
    assert one["counters"][0] == 1001 and two["max_rank"] == 1.0
    assert two["shard0"] == [5, 501]


def test_shard_partition_is_exact():
    from ldpc_decoders_amd.dist import Comm

    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 64, 65537):
            spans = [Comm(r, world).shard(100, total) for r in range(world)]
            assert sum(c for _, c in spans) == total
            pos = 100
            for s, c in spans:
                assert s == pos
                pos += c


# ---------------------------------------------------------------------------------------------------------------------------------
# The N > 1 DRIVER LAYER of bench.py (run_bench: DeviceSimulator rounds, frame shards, pipelining, the per-step all-reduce, the JSON
# line) on EIGHT gloo ranks.  The decoder handle is injected: here a CPU stand-in that draws the device's Philox noise (oracle model)
# and decodes with the C oracle -- exactly the role tests/ may give the oracle; bench.py's own main() always builds the HIP handle.
BENCH_WORKER = r'''
import json, os, sys
sys.path[:0] = [%(root)r, %(root)r + "/oracle", %(root)r + "/tests"]
import numpy as np
import torch
import bp_oracle as O, c_oracle as C
import bench
from ldpc_decoders_amd import dist


class OracleHandle:
    """Stand-in for _device.DecoderHandle: same simulate() contract, frames decoded by the CPU oracle."""
    log = []

    def __init__(self, code, alg, precision, backend):
        self.code, self.alg = code, alg
        self.g = O.Edges(code.m, code.n, code.edge_chk, code.edge_var)

    def simulate(self, channel, param, codeword, seed, stream_id, frame0, B, max_iter, counters, flags=0, hist_bins=0):
        assert channel == "biawgn" and codeword == 0
        OracleHandle.log.append((int(stream_id), int(frame0), int(B)))
        var = O.biawgn_noise_var(param)
        pri = np.stack([-2 * (-1 + np.sqrt(var) * O.device_biawgn_noise(seed, stream_id, f, self.g.n)) / var for f in range(frame0, frame0 + B)])
        xh, it = C.bp_decode(self.g, self.alg, None, pri, max_iter, nthreads=1)
        err = (xh != 0).sum(axis=1)
        counters[0] += B
        counters[1] += int((err > 0).sum())
        counters[2] += int(err.sum())
        counters[3] += int(it.sum())
        if hist_bins:
            counters[4:4 + hist_bins] += torch.from_numpy(np.bincount(np.minimum(it, hist_bins - 1), minlength=hist_bins))

    def last_stats(self):
        return "oracle", 0

    def kernel_name(self, simulate=False):
        return ""


batch_flags = %(batch_flags)s  # weak: --batch (768 / N per rank); strong: --total-batch (one fixed total, split by Comm.shard)
args = bench.parse_args(["--gpus", os.environ["WORLD_SIZE"], "--steps", "3", "--warmup", "1", "--repeats", "2"] + batch_flags +
                        ["--code", "7_4_hamming", "--snr", "2.0", "--max-iter", "10", "--points", "--no-profile", "--cpu-baseline-seconds", "0.2"])
# as bench.main(): rank 0 times the CPU baselines BEFORE it joins the process group (the other ranks wait at the rendezvous)
alg, channel, param = bench.resolve_workload(args.decoder, args.channel, args.param, args.snr)
cpu_base = bench.cpu_baseline(bench.load_code(args.code), alg, channel, param, args.max_iter, args.precision, args.cpu_baseline_seconds) if os.environ["RANK"] == "0" else None
comm = dist.init_from_env(prefer_gpu=False)
out = bench.run_bench(args, comm, make_handle=OracleHandle, device="cpu", cpu_base=cpu_base)
if out is not None:
    print(json.dumps(out))
json.dump(OracleHandle.log, open(sys.argv[1] + ".rank%%d" %% comm.rank, "w"))
dist.finalize()
'''


def _run_bench_layer(world, tmp_path, port, total_batch=None):
    flags = ["--total-batch", str(total_batch)] if total_batch is not None else ["--batch", str(768 // world)]
    code = BENCH_WORKER % {"root": ROOT, "batch_flags": repr(flags)}
    base = str(tmp_path / ("bench_w%d_%s" % (world, total_batch)))
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", code, base], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    lines = [ln for o in outs for ln in o.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]  # gloo's own connection banner
    logs = [json.load(open(base + ".rank%d" % r)) for r in range(world)]
    return lines, logs


@pytest.mark.timeout(900)
def test_bench_driver_layer_on_eight_ranks(tmp_path):
    lines1, logs1 = _run_bench_layer(1, tmp_path, 29621)
    lines8, logs8 = _run_bench_layer(8, tmp_path, 29622)
    assert len(lines1) == 1 and len(lines8) == 1                      # ONE JSON line, printed by rank 0 only
    one, eight = json.loads(lines1[0]), json.loads(lines8[0])
    assert one["n_gpus"] == 1 and eight["n_gpus"] == 8 and eight["scaling"] == "weak" and eight["steps"] == 3
    assert eight["config"]["batch_per_gpu"] == 96 and one["config"]["batch_per_gpu"] == 768
    # the same global frames were decoded: every counter of the whole job equals the single-rank run
    for k in ("frames_counted", "word_errors", "bit_errors", "mean_sweeps", "wer", "ber"):
        assert one[k] == eight[k], k
    assert eight["frames_counted"] == 3 * 768 and eight["timed_blocks"] == 2 and len(eight["blocks_ms_per_step"]) == 2
    assert eight["value"] > 0 and eight["ms_per_step_min"] <= eight["ms_per_step"] <= eight["ms_per_step_max"]
    # what makes an N > 1 line gradeable: the CPU baseline of rank 0's host, the roofline object and proof that the collective saw N ranks
    for line, world in ((one, 1), (eight, 8)):
        assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
        assert line["cpu_baseline"]["scipy"].get("value", 0) > 0, line["cpu_baseline"]["scipy"]
        assert isinstance(line["roofline"], dict) and "frac" in line["roofline"] and "bound" in line["roofline"]
        assert line["collective"]["ranks_seen"] == world
    assert eight["collective"]["backend"] == "gloo" and one["collective"]["backend"] is None
    # shard coverage: in every round the eight shards tile the round's frame range exactly, in rank order
    rounds = len(logs8[0])
    assert rounds == 1 + 2 * 3 and all(len(lg) == rounds for lg in logs8)
    for i in range(rounds):
        pos = logs1[0][i][1]
        assert logs1[0][i][2] == 768
        for r in range(8):
            stream, start, cnt = logs8[r][i]
            assert stream == logs1[0][i][0] and start == pos and cnt == 96
            pos += cnt
        assert pos == logs1[0][i][1] + 768


@pytest.mark.timeout(900)
def test_bench_strong_scaling_split_on_1_2_4_8_ranks(tmp_path):
    """`bench.py --total-batch T` (BASELINE configs 4 and 5 state their batch for the whole 8-GPU node): ONE fixed total per step, each rank
    its shard -- T = 763 is divisible by none of 2, 4, 8.  Whole-job counters must be identical for every N; the shards tile every round."""
    T = 763
    got = {}
    for i, world in enumerate((1, 2, 4, 8)):
        lines, logs = _run_bench_layer(world, tmp_path, 29631 + i, total_batch=T)
        assert len(lines) == 1
        got[world] = (json.loads(lines[0]), logs)
    one = got[1][0]
    assert one["frames_counted"] == 3 * T and one["scaling"] == "strong" and one["config"]["total_batch"] == T
    for world in (2, 4, 8):
        line, logs = got[world]
        assert line["n_gpus"] == world and line["scaling"] == "strong" and line["config"]["total_batch"] == T
        assert line["config"]["batch_per_gpu"] == -(-T // world)   # rank 0 holds the largest shard
        for k in ("frames_counted", "word_errors", "bit_errors", "mean_sweeps", "wer", "ber"):
            assert one[k] == line[k], (world, k)
        assert line["collective"]["ranks_seen"] == world and line["cpu_baseline"]["value"] > 0
        rounds = len(logs[0])
        assert rounds == 1 + 2 * 3
        for i in range(rounds):
            stream1, start1, cnt1 = got[1][1][0][i]
            assert cnt1 == T
            pos = start1
            for r in range(world):
                stream, start, cnt = logs[r][i]
                assert stream == stream1 and start == pos and cnt in (T // world, T // world + 1)
                pos += cnt
            assert pos == start1 + T


# ---------------------------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` started as a PLAIN process (the way the round driver starts --gpus 1): bench.self_launch must start the N
# ranks itself (a fresh torch.distributed.run child, the parent touching no GPU), relay rank 0's line, and never print a line for another N.
ENTRY = os.path.join(ROOT, "tests", "bench_cpu_entry.py")
TINY = ["--steps", "3", "--warmup", "1", "--repeats", "2", "--code", "7_4_hamming", "--snr", "2.0", "--max-iter", "10", "--points", "--no-profile",
        "--cpu-baseline-seconds", "0.2"]


def _plain(argv, env_extra=None, script=ENTRY):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, script] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    return p.returncode, [ln for ln in p.stdout.splitlines() if ln.strip()], p.stderr


@pytest.mark.timeout(900)
def test_plain_bench_command_launches_its_own_ranks():
    rc1, out1, _ = _plain(["--gpus", "1", "--batch", "192"] + TINY)
    rc2, out2, err2 = _plain(["--gpus", "2", "--batch", "96"] + TINY)
    assert rc1 == 0 and rc2 == 0, err2
    assert len(out1) == 1 and len(out2) == 1, out2             # stdout of the parent = rank 0's JSON line and nothing else
    one, two = json.loads(out1[0]), json.loads(out2[0])
    assert two["n_gpus"] == 2 and two["collective"]["ranks_seen"] == 2 and two["collective"]["backend"] == "gloo"
    assert one["n_gpus"] == 1 and one["collective"]["ranks_seen"] == 1
    assert "torch.distributed.run" in err2 and "--nproc-per-node 2" in err2
    for k in ("frames_counted", "word_errors", "bit_errors", "mean_sweeps", "wer", "ber"):   # same global frames, same counters
        assert one[k] == two[k], k
    assert two["cpu_baseline"]["value"] > 0 and isinstance(two["roofline"], dict)


def test_plain_bench_command_never_prints_a_line_for_another_n():
    # the real bench.py on a host without GPUs: --gpus 2 must end non-zero with no stdout at all (not an `n_gpus: 1` line)
    rc, out, err = _plain(["--gpus", "2", "--steps", "1"], script=os.path.join(ROOT, "bench.py"))
    import torch

    if torch.cuda.device_count() < 2:
        assert rc == 3 and out == [] and "requested" in err
    # a WORLD_SIZE that is not --gpus is refused the same way
    rc, out, err = _plain(["--gpus", "4", "--steps", "1"], env_extra={"WORLD_SIZE": "1", "RANK": "0"}, script=os.path.join(ROOT, "bench.py"))
    assert rc == 3 and out == [] and "WORLD_SIZE=1" in err


@pytest.mark.timeout(900)
def test_default_line_carries_the_baseline_configs_block():
    """The `baseline_configs` block of the default command (BASELINE configs 3-5 in bench.py; three tiny stand-ins here): every entry runs
    through the same run_bench on every rank, carries its own roofline object, and -- at N = 1 only -- its own CPU-port baseline."""
    env = {"BENCH_TEST_BASELINE_CONFIGS": "1"}
    rc1, out1, err1 = _plain(["--gpus", "1", "--batch", "96"] + TINY, env)
    rc2, out2, err2 = _plain(["--gpus", "2", "--batch", "48"] + TINY, env)
    assert rc1 == 0 and rc2 == 0, err1 + err2
    one, two = json.loads(out1[0]), json.loads(out2[0])
    for name, frames in (("tiny_spa", 2 * 64), ("tiny_msa", 3 * 48), ("tiny_f64", 2 * 32)):
        a, b = one["baseline_configs"][name], two["baseline_configs"][name]
        assert "error" not in a and "error" not in b, (a, b)
        assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and a["frames_counted"] == frames and b["frames_counted"] == 2 * frames  # weak scaling
        assert a["frames_per_s"] > 0 and b["frames_per_s"] > 0 and isinstance(a["roofline"], dict) and "frac" in a["roofline"]
        assert a["cpu_baseline"]["kind"] == "port" and a["cpu_baseline"]["value"] > 0 and "skipped" in a["cpu_baseline"]["scipy"]
        assert b["cpu_baseline"] is None                          # contract: the CPU baseline is timed at N = 1 only
        assert a["flags"].startswith("--code") and a["workload"]


def test_baseline_configs_table_names_the_baseline_configs():
    """The real table: one entry per BASELINE.json config 3-5 selector, flags parse, and only the driver's own command carries the block."""
    import bench

    names = [n for n, _, _ in bench.BASELINE_CONFIGS]
    assert names == ["config3_spa_bsc", "config3_bec", "config4", "config5"]
    d = bench.parse_args(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert bench.is_default_workload(d)
    for flags in (["--precision", "f32"], ["--decoder", "SPA"], ["--code", "gen:irg:10000"], ["--no-profile"], ["--no-baseline-configs"], ["--batch", "1024"]):
        assert not bench.is_default_workload(bench.parse_args(flags)), flags
    got = {}
    for name, _, flags in bench.BASELINE_CONFIGS:
        sub = bench.baseline_config_args(d, flags)
        got[name] = (bench.resolve_workload(sub.decoder, sub.channel, sub.param, sub.snr), sub.code, sub.batch, sub.precision, sub.points, sub.headline_only)
    assert got["config3_spa_bsc"] == (("SPA", "bsc", 0.07), "1200_3_6_rand_ldpc_1", 65536, "f32", [], True)
    assert got["config3_bec"][0] == ("BEC", "bec", 0.40) and got["config3_bec"][2] == 65536
    assert got["config4"][:3] == (("MSA", "biawgn", 1.2), "gen:irg:10000", 131072)
    assert got["config5"][:3] == (("MSA", "biawgn", 2.0), "gen:reg:64800:3:6", 32768)
