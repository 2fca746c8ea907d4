#!/bin/bash
# Runs ON THE GPU BOX: one gradeable bench line per BASELINE configuration (roofline AND cpu_baseline on each), at the BASELINE batch
# sizes (configs 4 / 5: the per-GPU shard of the 8-GPU batch).   tools/r05_lines.sh <tag> [cpu seconds per baseline leg]
TAG=${1:-r05}; CPU_S=${2:-5}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
run() { NAME=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.err; echo "$NAME rc=$?"; cp $OUT/bench_$NAME.json profiles/${TAG}_bench_$NAME.json 2>/dev/null; }
run config3_spa_bsc --decoder SPA --channel bsc --param 0.07 --precision f32 --batch 65536 --steps 20 --warmup 3 --cpu-baseline-seconds $CPU_S
run config3_bec     --decoder SPA --channel bec --param 0.40 --batch 65536 --steps 64 --warmup 8 --cpu-baseline-seconds $CPU_S
run config4 --code gen:irg:10000 --batch 131072 --snr 1.2 --steps 3 --warmup 1 --repeats 3 --precision f32 --points 1.8 --cpu-baseline-seconds $CPU_S
run config4_f64_stream --code gen:irg:10000 --batch 32768 --snr 1.8 --steps 2 --warmup 1 --repeats 3 --precision f64 --points --cpu-baseline-seconds $CPU_S
run config5 --code gen:reg:64800:3:6 --batch 32768 --snr 2.0 --steps 2 --warmup 1 --repeats 3 --precision f32 --points 1.0 --cpu-baseline-seconds $CPU_S
python - <<PY
import json
for f in ("config3_spa_bsc","config3_bec","config4","config4_f64_stream","config5"):
    try:
        d=json.load(open("$OUT/bench_%s.json" % f)); r=d["roofline"] or {}; c=d["cpu_baseline"] or {}
        print(f, "%.4g frames/s" % d["value"], d["ms_per_step"], "| roofline", r.get("bound"), r.get("binding_unit"), r.get("frac"), r.get("kernel"), "| cpu", c.get("value"), c.get("cores"), "| side", d.get("side_kernels_ms_per_step"), "host", d.get("host_overhead_ms_per_step"))
    except Exception as e: print(f, "FAILED", e, open("$OUT/bench_%s.err" % f).read()[-600:])
PY
