"""Long annealing runs of the fused backend's LDS layout planner, stored as plan files (csrc/ldpc_layout.hpp, "plan store").

    python tools/plan_codes.py --moves 300000000 --out ldpc_decoders_amd/plans [--codes-dir DIR] [--codes NAME ...] [--jobs 6]

Host-only: the planner runs behind `ldpc_plan_layout` (include/ldpc_hip.h), which needs no GPU.  For every code the plans of the
shapes a user meets are produced: fp32 (one plan serves min-sum, sum-product and the erasure decoder), fp64 min-sum, fp64
sum-product (edge order fixed).  One process per (code, shape); plans already present in --out are kept unless --force.
`gen:reg:<n>:<l>:<r>` / `gen:irg:<n>` name the generated ensembles of bench.py."""
import argparse
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("MSA", "f32"), ("MSA", "f64"), ("SPA", "f64")]

CHILD = r"""
import ctypes, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
from ldpc_decoders_amd import _lib, codes
from bench import load_code
name, alg, prec = %(code)r, %(alg)r, %(prec)r
code = load_code(name) if name.startswith("gen:") else codes.load_parity_mtx(os.path.join(%(dir)r, name + ".txt"))
lib = _lib.load()
info = (ctypes.c_double * 4)()
chk = np.ascontiguousarray(code.edge_chk, dtype=np.int32)
var = np.ascontiguousarray(code.edge_var, dtype=np.int32)
t0 = time.time()
_lib.check(lib.ldpc_plan_layout(code.m, code.n, code.E, chk.ctypes.data, var.ctypes.data, {"MSA": 0, "SPA": 1, "BEC": 2}[alg],
                                {"f32": 0, "f64": 1}[prec], %(moves)d, %(out)r.encode(), info))
print("%%-28s %%s %%s nw=%%d  gathers %%d cycles, conflicts %%d -> %%d  (%%.0f s)" %% (name, alg, prec, info[0], info[1], info[2], info[3], time.time() - t0), flush=True)
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--moves", type=int, default=300000000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "plans"))
    ap.add_argument("--codes", nargs="*", default=None, help="default: every *.txt of --codes-dir")
    ap.add_argument("--codes-dir", default=os.path.join(ROOT, "ldpc_decoders_amd", "data", "codes"))
    ap.add_argument("--jobs", type=int, default=max(1, (os.cpu_count() or 2) - 2))
    ap.add_argument("--nw", default="", help="LDPC_FUSED_NW: plan a non-default number of waves per frame")
    ap.add_argument("--shapes", nargs="*", default=None, help="subset of MSA:f32 MSA:f64 SPA:f64 (default: all three)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    names = a.codes if a.codes is not None else sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(a.codes_dir, "*.txt")))
    env = dict(os.environ)
    if a.nw:
        env["LDPC_FUSED_NW"] = a.nw
    shapes = SHAPES if not a.shapes else [tuple(x.split(":")) for x in a.shapes]
    todo = [(n, alg, prec) for n in names for alg, prec in shapes]
    running, rc = [], 0
    while todo or running:
        while todo and len(running) < a.jobs:
            n, alg, prec = todo.pop(0)
            src = CHILD % dict(root=ROOT, code=n, alg=alg, prec=prec, dir=a.codes_dir, moves=a.moves, out=a.out)
            running.append(subprocess.Popen([sys.executable, "-c", src], env=env))
        rc |= running.pop(0).wait()
    print(len(os.listdir(a.out)), "files in", a.out)
    return rc


if __name__ == "__main__":
    sys.exit(main())
