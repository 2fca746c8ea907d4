#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --kernel-trace of one config-5 step (n = 64 800, 32 768 frames, 2.0 dB) -> gpurun_out/<tag>/c5_timeline.txt, the
# kernel timeline of the LAST step (what profiles/rNN_config5_timeline.txt is made from).   tools/c5_profile.sh <tag>
TAG=${1:-r06t}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o k -- python3 $R/bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 2.0 --steps 2 --warmup 1 --repeats 1 --precision f32 --no-cpu-baseline --no-profile --points > $OUT/stats_c5.log 2>&1
cd $R
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/stats_c5/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size") or (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))))
rows.sort()
def short(n):
    n = n.replace("ldpc::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]
# the last step: from the last k_biawgn_tile launch on
starts = [i for i, r in enumerate(rows) if "k_biawgn_tile" in r[2]]
last = rows[starts[-1]:]
t0 = last[0][0]
with open("$OUT/c5_timeline.txt", "w") as fp:
    prev_end = t0
    for s, e, n, g in last:
        fp.write("%-32s t=%8.2f ms  dur=%7.3f ms  gap=%6.3f  grid=%s\n" % (short(n)[:32], (s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, g))
        prev_end = e
    fp.write("step: %.2f ms from the first kernel's start to the last kernel's end\n" % ((last[-1][1] - t0) / 1e6))
print(open("$OUT/c5_timeline.txt").read()[-2500:])
PY
find $OUT -name "*.db" -delete
