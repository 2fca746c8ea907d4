#!/bin/bash
# Runs ON THE GPU BOX: kernel-trace stats + LDS / VALU counters of the 16-wave fused kernel on the n = 10 000 irregular ensemble
# (BASELINE config 4).  Usage: tools/collect_config4.sh <tag>
set -u
TAG=${1:-r01c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG}_config4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/bench.py --code gen:irg:10000 --batch 131072 --snr 1.2 --steps 3 --warmup 1 --no-cpu-baseline --points > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_SQ -o p -- python3 $R/tools/prof_fused.py --code gen:irg:10000 --batch 16384 --snr 1.2 --reps 1 > $OUT/pmc_SQ.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_SQ2 -o p -- python3 $R/tools/prof_fused.py --code gen:irg:10000 --batch 16384 --snr 1.2 --reps 1 > $OUT/pmc_SQ2.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- python3 $R/tools/prof_fused.py --code gen:irg:10000 --batch 16384 --snr 1.2 --reps 1 > $OUT/pmc_$C.log 2>&1
done
find $OUT -name "*.db" -delete
ls $OUT
