#!/bin/bash
# Runs ON THE GPU BOX: everything a round's measurements are quoted from, under gpurun_out/<tag>/.   tools/round_measure.sh <tag> [quick]
# Order matters: the PMC counters are collected first and condensed ON THE BOX (profiles/roofline_counters.json), because bench.py and
# tools/measure_configs.py read them; whatever lands in profiles/ there is copied to gpurun_out/<tag>/profiles/ for the trip home.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -o faulthandler_timeout=300 > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_codes.py -m gpu -q -s -k "spa or config3 or soft" 2>/dev/null | grep -o -E "(fp64 sum-product|fp32 sum-product|config 3|soft LLR).*" > $OUT/parity_measured.txt
timeout 1500 bash tools/collect_rooflines.sh $TAG > $OUT/collect.log 2>&1
python tools/summarize_rooflines.py $TAG > $OUT/summarize.log 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python bench.py --precision f32 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err
timeout 600 python bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 1.0 --steps 2 --warmup 1 --repeats 3 --precision f32 --no-cpu-baseline --points > $OUT/bench_config5_n64800_1.0dB.json 2> $OUT/bench_c5a.err
timeout 600 python bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 2.0 --steps 2 --warmup 1 --repeats 3 --precision f32 --no-cpu-baseline --points > $OUT/bench_config5_n64800_2.0dB.json 2> $OUT/bench_c5b.err
timeout 600 python bench.py --code gen:irg:10000 --batch 131072 --snr 1.2 --steps 3 --warmup 1 --repeats 3 --precision f32 --no-cpu-baseline --points > $OUT/bench_config4_n10000_irregular.json 2> $OUT/bench_c4a.err
timeout 600 python bench.py --code gen:irg:10000 --batch 32768 --snr 1.8 --steps 2 --warmup 1 --repeats 3 --precision f64 --no-cpu-baseline --points > $OUT/bench_config4_n10000_irregular_f64_stream.json 2> $OUT/bench_c4b.err
# the driver's own command (BENCH_rNN.json): --steps 20 --warmup 5
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_drv.err
for f in bench bench_f32 bench_config5_n64800_1.0dB bench_config5_n64800_2.0dB bench_config4_n10000_irregular bench_config4_n10000_irregular_f64_stream bench_driver_command; do cp $OUT/$f.json profiles/${TAG}_$f.json 2>/dev/null; done
# rocprofv3 --kernel-trace --stats of the bench command itself (the contract's "same command"): its average kernel duration is what
# roofline.avg_launch_ms (HIP events inside bench.py) must agree with
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_f64 -o k -- python3 $R/bench.py --steps 40 --warmup 3 --repeats 2 --no-cpu-baseline --no-profile --points > $OUT/stats_bench_f64.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_f32 -o k -- python3 $R/bench.py --steps 40 --warmup 3 --repeats 2 --no-cpu-baseline --no-profile --points --precision f32 > $OUT/stats_bench_f32.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench_config5 -o k -- python3 $R/bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 1.0 --steps 2 --warmup 1 --repeats 1 --precision f32 --no-cpu-baseline --no-profile --points > $OUT/stats_bench_config5.log 2>&1 )
for v in bench_f64 bench_f32 bench_config5; do cp $OUT/stats_$v/k_kernel_stats.csv profiles/${TAG}_kernel_stats_$v.csv 2>/dev/null; done
timeout 900 python tools/measure_configs.py $TAG > $OUT/all_configs.log 2>&1
if [ "${2:-}" != "quick" ]; then
  ( time python tools/compare_curves.py --out $OUT/curves_vs_reference.md ) > $OUT/curves.log 2>&1
  ( time python tools/compare_curves.py --precision f32 --out $OUT/curves_vs_reference_fp32.md ) > $OUT/curves_f32.log 2>&1
  python tools/admm_rate.py > $OUT/admm_rate.log 2>&1
  python tools/ml_rate.py > $OUT/ml_rate.log 2>&1
  python tools/repack_probe.py 2.0 8192 > $OUT/repack_probe.log 2>&1
  python tools/repack_probe.py 1.8 32768 gen:irg:10000 >> $OUT/repack_probe.log 2>&1
fi
mkdir -p $OUT/profiles && cp profiles/${TAG}_* profiles/roofline_counters.json $OUT/profiles/ 2>/dev/null
find $OUT -name "*.db" -delete
tail -3 $OUT/gputest.log; cat $OUT/parity_measured.txt | tail -12
python - <<PY
import json
for f in ("bench.json","bench_f32.json","bench_config5_n64800_1.0dB.json","bench_config5_n64800_2.0dB.json","bench_config4_n10000_irregular.json","bench_config4_n10000_irregular_f64_stream.json"):
    try:
        d=json.load(open("$OUT/"+f)); r=d["roofline"] or {}
        print(f, "%.4g frames/s" % d["value"], d["ms_per_step"], (d["ms_per_step_min"], d["ms_per_step_max"]), "roofline", r.get("bound"), r.get("frac"), r.get("kernel"), "host", d.get("host_overhead_ms_per_step"), "side", d.get("side_kernels_ms_per_step"))
    except Exception as e: print(f, "FAILED", e)
PY
