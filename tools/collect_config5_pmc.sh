#!/bin/bash
# Runs ON THE GPU BOX: HBM byte counters (separate --pmc passes) of the streaming kernels on the (3,6) n = 64 800 shape, 8 192 frames, 1.0 dB
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_c5_$C -o p -- python3 $R/tools/prof_fused.py --backend stream --precision f32 --code gen:reg:64800:3:6 --batch 8192 --snr 1.0 --reps 1 > $OUT/pmc_c5_$C.log 2>&1
done
find $OUT -name "*.db" -delete
ls $OUT
