"""Shared helpers for the parity tests (fixtures are data captured from the reference by oracle/make_goldens.py)."""
import glob
import json
import os

import numpy as np

import bp_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# the reference's data/codes files ship inside the package (ldpc_decoders_amd/data/codes)
CODES_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ldpc_decoders_amd", "data", "codes")


def golden_edges(name):
    z = np.load(os.path.join(GOLDEN, "codes_edges.npz"))
    m, n = (int(v) for v in z[name + "__shape"])
    return O.Edges(m, n, z[name + "__chk"], z[name + "__var"])


def decode_cases(pattern="*"):
    return sorted(glob.glob(os.path.join(GOLDEN, "decode_%s.npz" % pattern)))


def load_case(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    for k in ("channel", "decoder", "code"):
        d[k] = str(d[k])
    for k in ("param",):
        d[k] = float(d[k])
    for k in ("codeword", "max_iter", "seed", "nframes"):
        d[k] = int(d[k])
    return d


def case_id(path):
    return os.path.basename(path)[len("decode_"):-4]


def kat_cases():
    with open(os.path.join(GOLDEN, "kat.json")) as fp:
        return json.load(fp)


def main_counter_cases():
    with open(os.path.join(GOLDEN, "main_counters.json")) as fp:
        return json.load(fp)


def expected_xhat(case):
    """Reference x_hat as float [B,n]; rows listed in raw_rows (iteration-0 return of the raw BI-AWGN word) hold y."""
    x = case["xhat"].astype(np.float64)
    for r in case["raw_rows"]:
        x[r] = case["y"][r]
    return x
