#!/bin/bash
# Runs ON THE GPU BOX: bench lines + rocprofv3 kernel stats of BASELINE configs 4 and 5 at their per-GPU sizes.  tools/collect_configs45.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SNR in 1.0 2.0; do
  python3 $R/bench.py --code gen:reg:64800:3:6 --batch 32768 --precision f32 --snr $SNR --steps 3 --warmup 1 --points --no-cpu-baseline > $OUT/bench_config5_n64800_${SNR}dB.json 2> $OUT/bench_config5_${SNR}.err
done
python3 $R/bench.py --code gen:irg:10000 --batch 131072 --precision f32 --snr 1.2 --steps 5 --warmup 1 --points 1.8 --no-cpu-baseline > $OUT/bench_config4_n10000_irregular.json 2> $OUT/bench_config4.err
python3 $R/bench.py --code gen:irg:10000 --batch 32768 --precision f64 --snr 1.8 --steps 2 --warmup 1 --points --no-cpu-baseline > $OUT/bench_config4_n10000_irregular_f64_stream.json 2> $OUT/bench_config4_f64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_config5 -o k -- python3 $R/bench.py --code gen:reg:64800:3:6 --batch 32768 --precision f32 --snr 2.0 --steps 2 --warmup 1 --points --no-cpu-baseline --no-profile > $OUT/stats_config5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_config4 -o k -- python3 $R/bench.py --code gen:irg:10000 --batch 131072 --precision f32 --snr 1.2 --steps 3 --warmup 1 --points --no-cpu-baseline --no-profile > $OUT/stats_config4.log 2>&1
find $OUT -name "*.db" -delete
ls $OUT | grep config
