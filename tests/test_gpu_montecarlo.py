"""GPU tests of the Monte-Carlo driver layer (ldpc_decoders_amd.montecarlo / dist): the path main.py really takes, the pipelined
rounds, and RCCL itself on one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import golden_edges

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _handle(alg="MSA", prec="f32", backend="auto", code="1200_3_6_rand_ldpc_1"):
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges(code)
    return DecoderHandle(Code.from_edges(g.m, g.n, g.chk, g.var), alg, prec, backend)


@pytest.mark.parametrize("alg,prec,channel,param", [("MSA", "f32", "biawgn", 2.0), ("MSA", "f64", "biawgn", 2.0), ("SPA", "f32", "bsc", 0.06),
                                                     ("BEC", "f32", "bec", 0.40), ("MSA", "f64", "bsc", 0.04)])
def test_main_py_path_is_the_in_kernel_path(alg, prec, channel, param):
    # main.py builds its DeviceSimulator WITHOUT a histogram (hist_bins = 0).  That call must take the same in-kernel
    # channel -> decode -> count path that bench.py times (hist_bins = max_iter + 1), and both must equal the composition
    # channel kernel -> ldpc_decode -> ldpc_count_errors on the streaming backend (same precision => same bits).
    from ldpc_decoders_amd.montecarlo import DeviceSimulator

    outs = {}
    for name, backend, bins in (("main", "auto", 0), ("bench", "auto", 51), ("composed", "stream", 0)):
        h = _handle(alg, prec, backend)
        sim = DeviceSimulator(h, channel, 50, 0, 1234, hist_bins=bins)
        outs[name] = sim.run_round(param, 3, 100, 4096)[:4].tolist()
        if backend == "auto":
            assert h.fused_info()["waves_per_frame"] > 0
    assert outs["main"] == outs["bench"] == outs["composed"], outs
    assert outs["main"][0] == 4096 and outs["main"][3] > 4096


def test_pipelined_rounds_equal_the_synchronous_loop():
    from ldpc_decoders_amd.montecarlo import DeviceSimulator

    h = _handle()
    sim = DeviceSimulator(h, "biawgn", 50, 0, 99, hist_bins=51)
    got = sim.run_point(2.3, 1, min_wec=60, batch_per_rank=512)
    tot = np.zeros(55, dtype=np.int64)
    frame0 = 0
    while tot[1] < 60:
        tot += sim.run_round(2.3, 1, frame0, 512)
        frame0 += 512
    assert (got["tot"], got["wec"], got["bec"], got["iter_sum"]) == tuple(int(v) for v in tot[:4])
    assert got["hist"] == tot[4:].tolist() and got["tot"] >= 3 * 512  # several rounds, so the pipeline really overlapped


WORKER = r'''
import json, os, sys
sys.path[:0] = [%(root)r, %(root)r + "/tests", %(root)r + "/oracle"]
import torch
from ldpc_decoders_amd import dist
from ldpc_decoders_amd.montecarlo import DeviceSimulator
from test_gpu_montecarlo import _handle
comm = dist.init_from_env()
import torch.distributed as td
sim = DeviceSimulator(_handle(), "biawgn", 50, 0, 7, comm, hist_bins=51)
a = sim.run_point(2.2, 0, min_wec=40, batch_per_rank=4096)
t = torch.arange(5, dtype=torch.int64, device="cuda") * (1 << 40)
comm.all_reduce_sum(t)
json.dump({"backend": comm.backend, "group": comm.group, "initialized": td.is_initialized(), "td_backend": td.get_backend() if td.is_initialized() else None,
           "point": a, "big": t.cpu().tolist(), "max": comm.max_float(1.5)}, open(sys.argv[1], "w"))
dist.finalize()
'''


@pytest.mark.timeout(900)
def test_rccl_all_reduce_of_the_counters_on_one_gpu(tmp_path):
    # De-risks the multi-GPU run on a 1-GPU box: a process group of ONE rank on the "nccl" backend (= RCCL on ROCm), so the int64
    # SUM all-reduce of the device-resident counters, the async (stream-ordered) form used by the pipelined rounds, the MAX
    # reduce of bench.py's timing and the barrier all execute inside RCCL.  Counters must equal the run without a group.
    outs = {}
    for name, extra in (("plain", {}), ("rccl", {"LDPC_DIST_FORCE_GROUP": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                                                 "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29741"})):
        env = dict(os.environ, **extra)
        env.pop("LDPC_DIST_BACKEND", None)
        path = str(tmp_path / (name + ".json"))
        r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = json.load(open(path))
    assert outs["rccl"]["backend"] == "nccl" and outs["rccl"]["group"] and outs["rccl"]["td_backend"] == "nccl"
    assert not outs["plain"]["initialized"]
    assert outs["rccl"]["point"] == outs["plain"]["point"] and outs["rccl"]["point"]["wec"] >= 40
    assert outs["rccl"]["big"] == [i << 40 for i in range(5)] and outs["rccl"]["max"] == 1.5
