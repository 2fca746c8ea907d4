#!/bin/bash
# Runs ON THE GPU BOX: decode time of the headline batch against the number of resident frames per CU (LDPC_FUSED_WAVES),
# fp64 and fp32 LDS kernels -- tells whether a kernel is bound by residency (latency) or by the LDS pipe.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-occ}
mkdir -p $OUT
for P in f64 f32; do
  for W in 1 2 3 4 5 6 8; do
    [ $P = f64 ] && [ $W -gt 4 ] && continue
    echo -n "$P frames_per_cu=$W " >> $OUT/sweep.txt
    LDPC_FUSED_WAVES=$W python3 $R/tools/prof_fused.py --precision $P --reps 5 2>/dev/null | tail -1 >> $OUT/sweep.txt
  done
done
cat $OUT/sweep.txt
