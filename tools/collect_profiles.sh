#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the bench command + separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / LDS counters), for the default fp64 arithmetic and the fp32 mode, written under
# gpurun_out/<tag>/ ; summarise with tools/summarize_profile.py.   Usage: tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in f64 f32; do
  for BK in fused stream; do
    B=auto; [ $BK = stream ] && B=stream
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_${BK}_$P -o k -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --points --no-profile --precision $P --backend $B > $OUT/stats_${BK}_$P.log 2>&1
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${BK}_${P}_$C -o p -- python3 $R/tools/prof_fused.py --reps 1 --precision $P --backend $B > $OUT/pmc_${BK}_${P}_$C.log 2>&1
    done
  done
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_fused_${P}_SQ -o p -- python3 $R/tools/prof_fused.py --reps 1 --precision $P --info $OUT/pmc_fused_${P}_SQ.info.json > $OUT/pmc_fused_${P}_SQ.log 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_fused_${P}_SQ2 -o p -- python3 $R/tools/prof_fused.py --reps 1 --precision $P > $OUT/pmc_fused_${P}_SQ2.log 2>&1
  # the kernels of the bench step itself (fp32: the fused SIMULATE kernel; fp64: channel + decode + count kernels)
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_sim_${P}_$C -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --points --no-profile --precision $P > $OUT/pmc_sim_${P}_$C.log 2>&1
  done
done
python3 $R/bench.py --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
python3 $R/bench.py --steps 6 --warmup 2 --precision f32 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err
find $OUT -name "*.db" -delete
ls $OUT
