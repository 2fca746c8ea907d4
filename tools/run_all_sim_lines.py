"""Every argument line of the reference's experiment tables (simulations.py: HMG, MAR, REG_BAD, REG_ENS, IREG_ENS -- 159 main.py runs;
the LP lines are not built) through ldpc_decoders_amd.main, in one process, with the tables' own --min-wec, on the GPU.  What
`run_sims.sh SEQL <CASE>` does for a user of the reference, as one regression run: every run must finish and produce its result
file with the reference's name and keys; a parameter whose word-error rate is too low to collect min_wec errors within --max-frames
(the tables go down to WER ~1e-9, where the reference itself would run for years) stops at the cap and is reported.

    python tools/run_all_sim_lines.py [--cases HMG MAR ...] [--out DIR]      (codes: ldpc_decoders_amd/data/codes)"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldpc_decoders_amd import codes, main as M, simulations  # noqa: E402


def run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", nargs="*", default=sorted(simulations.CASES))
    ap.add_argument("--out", default=None)
    ap.add_argument("--max-frames", type=int, default=1 << 25, help="safety cap per parameter (a point that needs more is reported, not failed)")
    a = ap.parse_args()
    os.environ.setdefault(codes.file_codes_dir_string, os.path.join(ROOT, "ldpc_decoders_amd", "data", "codes"))
    out = a.out or tempfile.mkdtemp(prefix="simlines_")
    os.makedirs(out, exist_ok=True)
    rows, t_all = [], time.time()
    for case in a.cases:
        for ln in simulations.lines(case):
            argv = ln.split() + ["--data_dir", out, "--console", "--log-freq", "1e9", "--max-frames", str(a.max_frames)]
            t0 = time.time()
            try:
                res = M.main(argv)
                frames = sum(int(r["tot"]) for r in res.values())
                worst = min(int(r["wec"]) for r in res.values())
                want = int([x for x in ln.split() if x.startswith("--min-wec")][0].split("=")[1])
                capped = [p for p, r in res.items() if int(r["wec"]) < want]  # points whose word-error rate is below min_wec / max_frames
                ok = all(int(res[p]["tot"]) >= a.max_frames for p in capped)
                rows.append(dict(case=case, line=ln, ok=ok, seconds=round(time.time() - t0, 2), frames=frames, points=len(res), min_wec_reached=worst,
                                 points_stopped_at_max_frames=len(capped)))
            except BaseException as e:  # SystemExit included: a run that stops is a failure here
                rows.append(dict(case=case, line=ln, ok=False, seconds=round(time.time() - t0, 2), error="%s: %s" % (type(e).__name__, e)))
            print(json.dumps(rows[-1]), flush=True)
    files = sorted(f for f in os.listdir(out) if f.endswith(".json"))
    bad = [r for r in rows if not r["ok"]]
    print(json.dumps(dict(runs=len(rows), failed=len(bad), result_files=len(files), frames=sum(r.get("frames", 0) for r in rows),
                          seconds=round(time.time() - t_all, 1))))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(run())
