#!/bin/bash
# Runs ON THE GPU BOX: everything a round's measurements are quoted from, under gpurun_out/<tag>/.   tools/round_measure.sh <tag> [quick]
# Order matters: the PMC counters are collected first and condensed ON THE BOX (profiles/roofline_counters.json), because bench.py and
# tools/measure_configs.py read them; whatever lands in profiles/ there is copied to gpurun_out/<tag>/profiles/ for the trip home.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
# .git does not travel to the box: the HEAD the measurements belong to is written into csrc/BUILD_HEAD before the call (git rev-parse HEAD)
export LDPC_HEAD=$(cat $R/ldpc_decoders_amd/csrc/BUILD_HEAD 2>/dev/null)
timeout 900 python -m pytest tests -m gpu -x -q -o faulthandler_timeout=300 > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_codes.py -m gpu -q -s -k "spa or config3 or soft" 2>/dev/null | grep -o -E "(fp64 sum-product|fp32 sum-product|config 3|soft LLR).*" > $OUT/parity_measured.txt
timeout 2400 bash tools/collect_rooflines.sh $TAG > $OUT/collect.log 2>&1
python tools/summarize_rooflines.py $TAG > $OUT/summarize.log 2>&1
# one gradeable line per BASELINE configuration: roofline AND cpu_baseline on every one (5 s of CPU work per baseline leg on configs 3-5)
run() { NAME=$1; shift; timeout 900 python bench.py "$@" > $OUT/$NAME.json 2> $OUT/$NAME.err; cp $OUT/$NAME.json profiles/${TAG}_$NAME.json 2>/dev/null; }
run bench
run bench_f32 --precision f32 --cpu-baseline-seconds 5
run bench_config3_spa_bsc --decoder SPA --channel bsc --param 0.07 --precision f32 --batch 65536 --steps 20 --warmup 3 --cpu-baseline-seconds 5
run bench_config3_bec --decoder SPA --channel bec --param 0.40 --batch 65536 --steps 64 --warmup 8 --cpu-baseline-seconds 5
run bench_config4 --code gen:irg:10000 --batch 131072 --snr 1.2 --steps 3 --warmup 1 --repeats 3 --precision f32 --points 1.8 --cpu-baseline-seconds 5
run bench_config4_f64_stream --code gen:irg:10000 --batch 32768 --snr 1.8 --steps 2 --warmup 1 --repeats 3 --precision f64 --points --cpu-baseline-seconds 5
run bench_config5 --code gen:reg:64800:3:6 --batch 32768 --snr 2.0 --steps 2 --warmup 1 --repeats 3 --precision f32 --points 1.0 --cpu-baseline-seconds 5
# the same two configurations as BASELINE states them for the whole node, on the GPUs this box has (strong scaling harness, N = 1 here)
run bench_config4_total_batch --code gen:irg:10000 --total-batch 1048576 --snr 1.2 --steps 2 --warmup 1 --repeats 1 --precision f32 --points --no-cpu-baseline --no-profile
run bench_config5_total_batch --code gen:reg:64800:3:6 --total-batch 262144 --snr 2.0 --steps 1 --warmup 1 --repeats 1 --precision f32 --points --no-cpu-baseline --no-profile
# the SURVEY 8(f) decoders: ADMM (LDS-resident kernel) and ML, each with roofline + CPU baseline
run bench_admm --decoder ADMM --param 2.2 --max-iter 300 --batch 65536 --steps 3 --warmup 1 --repeats 3 --cpu-baseline-seconds 5
run bench_ml --decoder ML --code 7_4_hamming --param 2.0 --precision f32 --batch 16777216 --steps 8 --warmup 2 --repeats 3 --max-iter 0 --cpu-baseline-seconds 3
# the driver's own command (BENCH_rNN.json): --steps 20 --warmup 5 -- its line carries `baseline_configs` (configs 3-5)
run bench_driver_command --gpus 1 --steps 20 --warmup 5
# rocprofv3 --kernel-trace --stats of the bench command itself (the contract's "same command"): its average kernel duration is what
# roofline.avg_launch_ms (HIP events inside bench.py) must agree with
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_f64 -o k -- python3 $R/bench.py --steps 40 --warmup 3 --repeats 2 --no-cpu-baseline --no-profile --points > $OUT/stats_bench_f64.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_f32 -o k -- python3 $R/bench.py --steps 40 --warmup 3 --repeats 2 --no-cpu-baseline --no-profile --points --precision f32 > $OUT/stats_bench_f32.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_config3_spa_bsc -o k -- python3 $R/bench.py --decoder SPA --channel bsc --param 0.07 --precision f32 --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-profile --points > $OUT/stats_bench_c3spa.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_config3_bec -o k -- python3 $R/bench.py --decoder SPA --channel bec --param 0.40 --steps 64 --warmup 8 --repeats 2 --no-cpu-baseline --no-profile --points > $OUT/stats_bench_c3bec.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_config5 -o k -- python3 $R/bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 2.0 --steps 2 --warmup 1 --repeats 1 --precision f32 --no-cpu-baseline --no-profile --points > $OUT/stats_bench_config5.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_config4 -o k -- python3 $R/bench.py --code gen:irg:10000 --batch 131072 --snr 1.2 --steps 3 --warmup 1 --repeats 1 --precision f32 --no-cpu-baseline --no-profile --points > $OUT/stats_bench_config4.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_admm -o k -- python3 $R/bench.py --decoder ADMM --param 2.2 --max-iter 300 --batch 65536 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-profile > $OUT/stats_bench_admm.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_ml -o k -- python3 $R/bench.py --decoder ML --code 7_4_hamming --param 2.0 --precision f32 --batch 16777216 --steps 8 --warmup 2 --repeats 1 --max-iter 0 --no-cpu-baseline --no-profile > $OUT/stats_bench_ml.log 2>&1 )
for v in bench_f64 bench_f32 bench_config3_spa_bsc bench_config3_bec bench_config4 bench_config5 bench_admm bench_ml; do cp $OUT/stats_$v/k_kernel_stats.csv profiles/${TAG}_kernel_stats_$v.csv 2>/dev/null; done
timeout 900 python tools/measure_configs.py $TAG > $OUT/all_configs.log 2>&1
python tools/host_rate.py > $OUT/host_rate.log 2>&1
if [ "${2:-}" != "quick" ]; then
  ( time python tools/compare_curves.py --out $OUT/curves_vs_reference.md ) > $OUT/curves.log 2>&1
  ( time python tools/compare_curves.py --precision f32 --out $OUT/curves_vs_reference_fp32.md ) > $OUT/curves_f32.log 2>&1
  python tools/admm_rate.py > $OUT/admm_rate.log 2>&1
  python tools/ml_rate.py > $OUT/ml_rate.log 2>&1
fi
mkdir -p $OUT/profiles && cp profiles/${TAG}_* profiles/roofline_counters.json $OUT/profiles/ 2>/dev/null
find $OUT -name "*.db" -delete
tail -3 $OUT/gputest.log; cat $OUT/parity_measured.txt | tail -12; cat $OUT/host_rate.log
python - <<PY
import json
for f in ("bench","bench_f32","bench_config3_spa_bsc","bench_config3_bec","bench_config4","bench_config4_f64_stream","bench_config5","bench_config4_total_batch","bench_config5_total_batch","bench_admm","bench_ml","bench_driver_command"):
    try:
        d=json.load(open("$OUT/"+f+".json")); r=d["roofline"] or {}; c=d.get("cpu_baseline") or {}
        print(f, "%.4g frames/s" % d["value"], d["ms_per_step"], (d["ms_per_step_min"], d["ms_per_step_max"]), "roofline", r.get("bound"), r.get("binding_unit"), r.get("frac"), r.get("kernel"), "cpu", c.get("value"), "host", d.get("host_overhead_ms_per_step"), "side", d.get("side_kernels_ms_per_step"))
    except Exception as e: print(f, "FAILED", e)
PY
