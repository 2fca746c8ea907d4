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
