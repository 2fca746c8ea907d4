"""Times the TRUE reference (imported from /root/reference, build container only) and the build's scipy.sparse baseline
(oracle/scipy_baseline.py) on identical frames, one host core, and writes tests/golden/reference_timing.json:

    per (decoder, SNR): frames, mean iterations, frames/s of both, and `calibration` = reference frames/s / baseline frames/s --
    the factor that turns a scipy-baseline rate measured on another host (bench.py's cpu_baseline leg on the GPU box) into an
    estimate of the reference's rate there.

    python oracle/make_timing.py [--frames 200]          (about five minutes)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]
import ref_import  # noqa: E402
from scipy_baseline import ScipyBP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    a = ap.parse_args()
    R = ref_import.load()
    code = R.codes.get_code("1200_3_6_rand_ldpc_1")
    chk, var = np.where(code.parity_mtx)
    m, n = code.parity_mtx.shape
    x = np.zeros(n, dtype=np.int64)
    out = {"host": "build container, 1 of %d cores (%s)" % (os.cpu_count(), open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")),
           "code": "1200_3_6_rand_ldpc_1", "max_iter": 50, "frames_per_point": a.frames,
           "note": "reference = thadikari/ldpc_decoders src/biawgn.py MSA/SPA .decode(y) imported unmodified; baseline = oracle/scipy_baseline.py; "
                   "identical received words (np.random.seed(99)), single thread", "points": []}
    for alg, snr in [("MSA", 1.0), ("MSA", 2.0), ("MSA", 3.0), ("SPA", 1.0), ("SPA", 2.0)]:
        chan = R.biawgn.Channel(snr)
        dec = getattr(R.biawgn, alg)(snr, code, max_iter=50)
        np.random.seed(99)
        ys = [chan.send(x) for _ in range(a.frames)]
        calls = {"n": 0}
        inner = dec.dec.decode_

        def counted(*args, _f=inner, **kw):
            calls["n"] += 1
            return _f(*args, **kw)

        dec.dec.decode_ = counted
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            ref_out = [np.asarray(dec.decode(y)) for y in ys]
        t_ref = time.perf_counter() - t0
        var_n = 10 ** (-snr / 10)
        base = ScipyBP(m, n, chk, var, alg, 50)
        its = 0
        t0 = time.perf_counter()
        base_out = []
        for y in ys:
            base_out.append(np.asarray(base.decode(y, -2 * y / var_n)))
            its += base.iterations
        t_base = time.perf_counter() - t0
        same = sum(int((p == q).all()) for p, q in zip(ref_out, base_out))
        pt = dict(decoder=alg, snr_db=snr, frames=a.frames, mean_iters=calls["n"] / a.frames, seconds=t_ref, frames_per_s=a.frames / t_ref,
                  baseline_seconds=t_base, baseline_frames_per_s=a.frames / t_base, baseline_mean_iters=its / a.frames,
                  calibration=(a.frames / t_ref) / (a.frames / t_base), identical_frames=same)
        out["points"].append(pt)
        print(json.dumps(pt), flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "reference_timing.json"), "w") as fp:
        json.dump(out, fp, indent=1)


if __name__ == "__main__":
    main()
