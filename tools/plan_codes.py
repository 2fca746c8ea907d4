"""Long annealing runs of the fused backend's LDS layout planner, stored as plan files (ldpc_layout.hpp, "plan store").

    python tools/plan_codes.py --moves 1000000000 --out gpurun_out/plans [--codes NAME ...]

Each code is planned in its own process (the planner is single-threaded host code inside fused_plan_create, which needs
a GPU only because the decoder handle allocates its tables there).  Copy the resulting <key>.plan files into
ldpc_decoders_amd/plans/ to ship them."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODES = ["1200_3_6_rand_ldpc_1", "1200_3_6_ldpc", "512_3_6_rand_ldpc_2", "1200_rho_x5_rand_ldpc_5", "margulis", "gen:irg:10000"]

CHILD = r"""
import os, sys, time
sys.path.insert(0, %(root)r)
from ldpc_decoders_amd import codes
from ldpc_decoders_amd._device import DecoderHandle
from bench import load_code
t0 = time.time()
code = load_code(%(code)r) if %(code)r.startswith("gen:") else codes.get_code(%(code)r)  # gen:reg:<n>:<l>:<r> / gen:irg:<n> as in bench.py
try:
    h = DecoderHandle(code, "MSA", os.environ.get("LDPC_PLAN_PRECISION", "f32"), "fused")
except Exception as e:  # no fused shape for this (code, LDPC_FUSED_NW): nothing to plan
    print(%(code)r, "nw", os.environ.get("LDPC_FUSED_NW", "auto"), "skipped:", e, flush=True)
    sys.exit(0)
print(%(code)r, "nw", os.environ.get("LDPC_FUSED_NW", "auto"), "%%.0fs" %% (time.time() - t0), h.fused_info(), flush=True)
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--moves", type=int, default=1000000000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "plans"))
    ap.add_argument("--codes", nargs="*", default=CODES)
    ap.add_argument("--codes-dir", default=os.path.join(ROOT, "tests", "golden", "codes"))
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"], help="f64: the shapes of the fp64 min-sum kernel")
    ap.add_argument("--nw", nargs="*", default=["", "1"], help="LDPC_FUSED_NW values to plan for ('' = the default shape)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    procs = []
    for code in a.codes:
        for nw in a.nw:  # by default: the default shape and the one-wave-per-frame shape (LDPC_FUSED_NW=1)
            env = dict(os.environ, LDPC_FUSED_PLAN_MOVES=str(a.moves), LDPC_FUSED_PLAN_SAVE=a.out, LDPC_FUSED_LAYOUT="replan",
                       FILE_CODES_DIR=a.codes_dir, LDPC_PLAN_PRECISION=a.precision)
            if nw:
                env["LDPC_FUSED_NW"] = nw
            procs.append(subprocess.Popen([sys.executable, "-c", CHILD % dict(root=ROOT, code=code)], env=env))
    rc = 0
    for p in procs:
        rc |= p.wait()
    print(sorted(os.listdir(a.out)))
    return rc


if __name__ == "__main__":
    sys.exit(main())
