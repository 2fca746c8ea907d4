"""The layout-plan store under concurrent decoder construction (no GPU: the host-only entry point ldpc_plan_layout with moves < 0
does exactly what ldpc_decoder_create does -- store lookup, otherwise ONE process per node anneals and publishes the plan)."""
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes, json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
from ldpc_decoders_amd import _lib, codes
code = codes.rand_reg_ldpc(600, 3, 6, np.random.RandomState(5))   # not among the shipped plans
lib = _lib.load()
info = (ctypes.c_double * 4)()
chk = np.ascontiguousarray(code.edge_chk, dtype=np.int32); var = np.ascontiguousarray(code.edge_var, dtype=np.int32)
t0 = time.time()
_lib.check(lib.ldpc_plan_layout(code.m, code.n, code.E, chk.ctypes.data, var.ctypes.data, 0, 0, -1, sys.argv[1].encode(), info))
json.dump({"seconds": time.time() - t0, "info": list(info)}, open(sys.argv[2], "w"))
'''


@pytest.mark.timeout(600)
def test_one_process_per_node_anneals_the_others_load(tmp_path):
    store = tmp_path / "plans"
    env = dict(os.environ, LDPC_FUSED_PLAN_MOVES="3000000", LDPC_FUSED_PLAN_DIR=str(store), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}, str(store), str(tmp_path / ("r%d.json" % r))], env=env) for r in range(6)]
    for p in procs:
        assert p.wait(timeout=500) == 0
    outs = [json.load(open(tmp_path / ("r%d.json" % r))) for r in range(6)]
    from_store = [o for o in outs if o["info"][0] < 0]
    annealed = [o for o in outs if o["info"][0] > 0]
    assert len(annealed) == 1 and len(from_store) == 5, outs       # exactly one owner of the lock
    assert len({tuple(o["info"][1:]) for o in outs}) == 1            # everybody ends with the same plan (cycles, conflicts before / after)
    files = sorted(os.listdir(store))
    assert len(files) == 1 and files[0].endswith(".plan"), files    # no lock or temporary file left behind
    assert oct(os.stat(store).st_mode & 0o777) == "0o700"           # directories the library creates are private
    # a later construction finds the file at once
    p = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, str(store), str(tmp_path / "late.json")], env=env, timeout=300)
    late = json.load(open(tmp_path / "late.json"))
    assert p.returncode == 0 and late["info"][0] < 0 and late["seconds"] < min(o["seconds"] for o in annealed) / 2 + 1.0


def test_stale_lock_is_ignored(tmp_path):
    store = tmp_path / "plans"
    os.makedirs(store)
    env = dict(os.environ, LDPC_FUSED_PLAN_MOVES="200000", LDPC_FUSED_PLAN_DIR=str(store), OMP_NUM_THREADS="1")
    # find the plan's file name, then leave an ancient lock behind as a crashed owner would
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, str(store), str(tmp_path / "a.json")], env=env, timeout=300)
    assert r.returncode == 0
    (name,) = os.listdir(store)
    os.unlink(store / name)
    lock = store / ("." + name + ".lock")
    lock.write_text("")
    old = time.time() - 3600
    os.utime(lock, (old, old))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, str(store), str(tmp_path / "b.json")], env=env, timeout=300)
    assert r.returncode == 0 and time.time() - t0 < 60
    assert sorted(os.listdir(store)) == [name]
