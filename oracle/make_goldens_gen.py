"""Fixture for the irregular-ensemble generator: the reference's own degree distributions and node counts.

TEST INFRASTRUCTURE (build container only; imports the reference through oracle/ref_import.py).  For rho(x) = x^4, x^5, x^6 at design
rate 1/2: lambda(x) from ldpc.solve_dist (src/ldpc.py:83-94) and, for n = 1200 and 10 000, the node counts int(L_i * n) that
gen_rand_irg_ldpc places before its hard-coded `extra` variables (src/ldpc.py:158-161) -> tests/golden/irregular_ensembles.json.
"""
import importlib
import json
import os

import numpy as np

import ref_import

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    ref_import.load()
    ldpc = importlib.import_module("ldpc")
    out = {}
    for rho in (4, 5, 6):
        dist = ldpc.solve_dist("rho_r=%d" % rho, .5, ldpc.reg_pol(rho))
        lam = [float(v) for v in dist.lambda_p]  # highest power first; the coefficient of x^(d-1) is the edge fraction of degree d
        L_p, _ = ldpc.gen_L_R(dist)              # node perspective, highest power first, constant term last
        entry = {"lambda_edge": {str(len(lam) - i): lam[i] for i in range(len(lam)) if abs(lam[i]) > 1e-12}, "node_counts": {}}
        for n in (1200, 10000):
            counts = [int(it * n) for it in L_p]
            deg = len(counts) - 1
            entry["node_counts"][str(n)] = {str(deg - i): c for i, c in enumerate(counts) if c}
        out[str(rho)] = entry
        print(rho, entry)
    with open(os.path.join(GOLD, "irregular_ensembles.json"), "w") as fp:
        json.dump(out, fp, indent=1)


if __name__ == "__main__":
    main()
