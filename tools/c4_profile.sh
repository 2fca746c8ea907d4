cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05x
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05x/stats_c4 -o k -- python3 $R/bench.py --code gen:irg:10000 --batch 32768 --snr 1.8 --steps 2 --warmup 1 --repeats 1 --precision f64 --no-cpu-baseline --no-profile --points > $R/gpurun_out/r05x/stats_c4.log 2>&1
head -16 $R/gpurun_out/r05x/stats_c4/k_kernel_stats.csv | cut -c1-200
find $R/gpurun_out/r05x -name "*.db" -delete
