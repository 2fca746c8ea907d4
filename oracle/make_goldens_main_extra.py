"""Append multi-parameter n=1200 runs of the reference's CLI driver to tests/golden/main_counters.json.

TEST INFRASTRUCTURE (build container only; imports the reference through oracle/ref_import.py).  The reference consumes
exactly n draws of the global np.random stream per frame (src/main.py:37-40), so every --params value after the first starts
where the previous one stopped: these runs pin that hand-over for the chunked --exact mode of ldpc_decoders_amd.montecarlo.
"""
import contextlib
import io
import json
import os
import runpy
import shutil
import sys

import numpy as np

import ref_import

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
RUNS = [
    (77, "biawgn 1200_3_6_rand_ldpc_1 MSA --codeword 0 --min-wec 5 --max-iter 50 --params 1.5 2.0 2.25"),
    (78, "bsc 1200_3_6_rand_ldpc_1 MSA --codeword 0 --min-wec 4 --max-iter 20 --params 0.06 0.05 0.045"),
    (79, "bec 1200_3_6_rand_ldpc_1 SPA --codeword 0 --min-wec 4 --max-iter 50 --params 0.42 0.4"),
]


def main():
    ref_import.load()
    path = os.path.join(GOLD, "main_counters.json")
    with open(path) as fp:
        out = json.load(fp)
    have = {(r["seed"], r["argline"]) for r in out}
    tmp = "/tmp/ldpc_goldens_main_extra"
    shutil.rmtree(tmp, ignore_errors=True)
    for seed, line in RUNS:
        if (seed, line) in have:
            continue
        old = sys.argv
        sys.argv = ["main.py"] + line.split() + ["--data_dir", tmp, "--console"]
        np.random.seed(seed)
        try:
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                runpy.run_path(os.path.join(ref_import.REF_ROOT, "src", "main.py"), run_name="__main__")
        finally:
            sys.argv = old
        newest = max(os.listdir(tmp), key=lambda f: os.path.getmtime(os.path.join(tmp, f)))
        with open(os.path.join(tmp, newest)) as fp:
            data = json.load(fp)
        out.append(dict(seed=seed, argline=line, file_name=newest, result=data))
        print("  main:", line, "->", {k: data[k] for k in ("tot", "wec", "bec")}, flush=True)
    with open(path, "w") as fp:
        json.dump(out, fp, indent=1)


if __name__ == "__main__":
    main()
