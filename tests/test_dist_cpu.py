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
