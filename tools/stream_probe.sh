#!/bin/bash
# Runs ON THE GPU BOX: the streaming backend's per-kernel HBM rates (HIP events) on the shapes it is used for.  tools/stream_probe.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/stream_probe}
mkdir -p $OUT
cd $R
python bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 1.0 --steps 2 --warmup 1 --precision f32 --backend stream --no-cpu-baseline --points > $OUT/c5_1.0dB.json 2> $OUT/c5_1.0dB.err
python bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 2.0 --steps 2 --warmup 1 --precision f32 --backend stream --no-cpu-baseline --points > $OUT/c5_2.0dB.json 2> $OUT/c5_2.0dB.err
python bench.py --code gen:irg:10000 --batch 32768 --snr 1.2 --steps 2 --warmup 1 --precision f32 --backend stream --no-cpu-baseline --points > $OUT/c4_1.2dB.json 2> $OUT/c4_1.2dB.err
python bench.py --code gen:irg:10000 --batch 32768 --snr 1.8 --steps 2 --warmup 1 --precision f64 --backend stream --no-cpu-baseline --points > $OUT/c4_1.8dB_f64.json 2> $OUT/c4_1.8dB_f64.err
python bench.py --batch 65536 --snr 1.0 --steps 3 --warmup 1 --precision f32 --backend stream --no-cpu-baseline --points > $OUT/c2_f32.json 2> $OUT/c2_f32.err
python bench.py --batch 65536 --snr 1.0 --steps 3 --warmup 1 --precision f64 --backend stream --no-cpu-baseline --points > $OUT/c2_f64.json 2> $OUT/c2_f64.err
for f in $OUT/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]
print(" value %.4g frames/s  ms/step %.2f  sweeps %.2f  alg %.0f GB/s  kernels %s  host_overhead %.2f" % (d["value"], d["ms_per_step"], d["mean_sweeps"], d["algorithmic_GBps"], r.get("all_kernels_ms"), d.get("host_overhead_ms_per_step") or -1))
print("  dominant", r["kernel"], r["achieved"], r["frac"])
PY
done
