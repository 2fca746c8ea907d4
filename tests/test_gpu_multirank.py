"""N>1 path of bench.py / the CLI on a 1-GPU box: two ranks share GPU 0 (LDPC_DIST_BACKEND=gloo), launched the way the
driver launches them (torch.distributed.run, --master-addr 127.0.0.1).  Whole-job counters must equal the 1-rank run over
the same global frame range (noise is keyed by global frame index)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(nproc, port, batch):
    env = dict(os.environ, LDPC_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
           "--batch", str(batch), "--snr", "2.0", "--points", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_two_ranks_equal_one_rank():
    one = _bench(1, 29721, 8192)   # 2 steps x 8192 frames on 1 rank
    two = _bench(2, 29722, 4096)   # 2 steps x 2 ranks x 4096 frames: same global frame ranges per step
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    for key in ("mean_sweeps", "wer", "ber"):
        assert one[key] == two[key], key
    assert two["value"] > 0 and two["roofline"]["kernel_class"] == "fused_decode" and two["roofline"]["kernel"].startswith("k_fused_f64<")
    assert two["timed_blocks"] == 5 and two["ms_per_step_min"] <= two["ms_per_step"] <= two["ms_per_step_max"]
    assert one["frames_counted"] == two["frames_counted"] == 2 * 8192 and one["bit_errors"] == two["bit_errors"]


def _bench_flags(nproc, port, flags):
    env = dict(os.environ, LDPC_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--no-cpu-baseline", "--points"] + flags
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_strong_split_on_the_streaming_backend_and_multi_round_launches_with_two_ranks():
    # (1) BASELINE configs 4 / 5 state their batch for the whole node: `--total-batch T` splits ONE fixed total over the ranks (strong
    # scaling).  Streaming backend (pipeline depth 1: the decode polls the host), T = 1001 is odd: the ranks hold 501 + 500 frames.
    flags = ["--steps", "2", "--warmup", "1", "--repeats", "1", "--total-batch", "1001", "--snr", "2.0", "--precision", "f32", "--backend", "stream"]
    one, two = _bench_flags(1, 29731, flags), _bench_flags(2, 29732, flags)
    assert one["scaling"] == two["scaling"] == "strong" and two["config"]["total_batch"] == 1001 and two["config"]["batch_per_gpu"] == 501
    assert one["config"]["backend"] == two["config"]["backend"] == "stream"
    for key in ("frames_counted", "word_errors", "bit_errors", "mean_sweeps", "wer", "ber"):
        assert one[key] == two[key], key
    assert one["frames_counted"] == 2 * 1001 and two["roofline"]["bound"] == "hbm"
    # (2) the erasure decoder sends up to 32 steps per launch (ldpc_simulate_rounds); with two ranks every step's frame range is still
    # split in rank order (round_stride = the whole job's frames per step) and each step keeps its own counter row
    flags = ["--steps", "37", "--warmup", "2", "--repeats", "1", "--decoder", "SPA", "--channel", "bec", "--param", "0.41", "--total-batch", "4097"]
    one, two = _bench_flags(1, 29733, flags), _bench_flags(2, 29734, flags)
    assert one["config"]["steps_per_launch"] == two["config"]["steps_per_launch"] == 32 and two["config"]["decoder"] == "BEC"
    for key in ("frames_counted", "word_errors", "bit_errors", "mean_sweeps", "wer", "ber"):
        assert one[key] == two[key], key
    assert one["frames_counted"] == 37 * 4097 and one["word_errors"] > 0


@pytest.mark.timeout(1800)
def test_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` as a plain process (no WORLD_SIZE -- the way the round driver starts --gpus 1): bench.self_launch starts the two
    ranks itself (a torch.distributed.run child; the parent makes no GPU call) and relays rank 0's line -- never an `n_gpus: 1` line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LDPC_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4096", "--snr", "2.0",
           "--points", "--no-cpu-baseline", "--no-baseline-configs"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["collective"]["ranks_seen"] == 2 and two["frames_counted"] == 2 * 8192
    assert two["roofline"]["kernel"].startswith("k_fused_f64<")
    # without the gloo override a 1-GPU box must refuse --gpus 2 with no line at all (RCCL needs one GPU per rank)
    import torch

    if torch.cuda.device_count() < 2:
        env.pop("LDPC_DIST_BACKEND")
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 3 and out.stdout.strip() == "" and "requested" in out.stderr
