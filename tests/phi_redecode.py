"""Worker of tests/test_gpu_large_codes.py::test_config3_spa_bsc_full_batch: re-decodes a batch with the fp64 phi-domain oracle
(oracle/bp_oracle.py) on all host cores.  Started as a FRESH process (python tests/phi_redecode.py in.npz out.npz code max_iter procs):
the test process has an initialised GPU runtime with live threads, and forking a worker pool from it can deadlock a child.
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))


def _chunk(task):
    import bp_oracle as O
    from helpers import golden_edges

    name, y, pri, max_iter = task
    return O.bp_decode(golden_edges(name), "SPA_PHI", y, pri, max_iter)


if __name__ == "__main__":
    src, dst, name, max_iter, procs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    d = np.load(src)
    y, pri = d["y"], d["pri"]
    with mp.get_context("fork").Pool(procs) as pool:
        parts = pool.map(_chunk, [(name, y[i:i + 256], pri[i:i + 256], max_iter) for i in range(0, len(y), 256)])
    np.savez(dst, x=np.concatenate([p[0] for p in parts]), it=np.concatenate([p[1] for p in parts]))
