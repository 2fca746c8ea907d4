#!/bin/bash
# Runs ON THE GPU BOX: the frame repack folded into the sweep behind it (k_repack_map + GATHER passes) against the separate copy kernel
# (LDPC_STREAM_REPACK_FOLD=0), same box, configs 5 (n = 64 800 fp32, 2.0 dB) and 4 (n = 10 000 fp64 streaming, 1.8 dB).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r06/ab_fold}
mkdir -p $OUT
for rep in 1 2; do
  for F in 1 0; do
    LDPC_STREAM_REPACK_FOLD=$F python $R/bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 2.0 --steps 2 --warmup 1 --repeats 3 --precision f32 --points --no-cpu-baseline > $OUT/c5_fold$F.$rep.json 2>/dev/null
    LDPC_STREAM_REPACK_FOLD=$F python $R/bench.py --code gen:irg:10000 --batch 32768 --snr 1.8 --steps 2 --warmup 1 --repeats 3 --precision f64 --points --no-cpu-baseline > $OUT/c4_f64_fold$F.$rep.json 2>/dev/null
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "%.5g frames/s" % d["value"], d["ms_per_step"], "side", d.get("side_kernels_ms_per_step"), "frac", d["roofline"].get("frac"), "wer", d["wer"], "sweeps", d["mean_sweeps"])
    except Exception as e: print(f, "FAILED", e)
PY
