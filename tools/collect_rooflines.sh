#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): PMC counters of the kernel each BASELINE configuration is dominated by, on the SIMULATE variants
# bench.py / main.py actually launch -- LDS / VALU counters of the LDS-resident kernels, HBM bytes of every kernel (separate
# FETCH_SIZE / WRITE_SIZE passes with a known-size calibration copy, as MI355X_MICROARCH.md prescribes) -- and a rocprofv3
# --kernel-trace --stats pass of the same commands.  Raw output under gpurun_out/<tag>/; tools/summarize_rooflines.py <tag> condenses
# it into profiles/<tag>_roofline_counters.json (+ profiles/roofline_counters.json, the copy bench.py and measure_configs.py read).
#   tools/collect_rooflines.sh <tag> [case ...]
set -u
TAG=${1:-r04}
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"
# LDS instruction split + queue counters (which unit of the LDS path binds: the array, or the address / data transfer of the stores)
SQ4="SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
# VALU instruction mix (issue-cycle model of the VALU-bound kernels): only the counters this rocprofv3 knows
SQ3=""
LIST=$(rocprofv3 -L 2>/dev/null)
for C3 in SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT; do
  if echo "$LIST" | grep -q "Counter_Name *:.*\b$C3\b"; then SQ3="$SQ3 $C3"; fi
done
echo "VALU mix counters:$SQ3"
# name | sim_driver arguments | passes
CASES=(
 "c2_f64|--code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f64 --launches 3|sq hbm"
 "c2_f32|--code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f32 --launches 3|sq hbm"
 "c2_f32_grid|--code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f32 --prior-grid 8 --launches 3|sq hbm"
 "c3_spa_bsc_f32|--code 1200_3_6_rand_ldpc_1 --alg SPA --channel bsc --param 0.07 --batch 65536 --precision f32 --launches 3|sq hbm"
 "c3_bec|--code 1200_3_6_rand_ldpc_1 --alg BEC --channel bec --param 0.40 --batch 1048576 --precision f32 --launches 3|sq hbm"
 "c3_spa_biawgn_f64|--code 1200_3_6_rand_ldpc_1 --alg SPA --channel biawgn --param 1.5 --batch 65536 --precision f64 --launches 2|sq hbm"
 "f_admm|--code 1200_3_6_rand_ldpc_1 --alg ADMM --channel biawgn --param 2.2 --batch 8192 --precision f64 --max-iter 300 --launches 2 --warm 2|sq"
 "f_ml|--code 7_4_hamming --alg ML --channel biawgn --param 2.0 --batch 16777216 --precision f32 --max-iter 0 --launches 3 --warm 2|sq"
 "c4_f32|--code gen:irg:10000 --alg MSA --channel biawgn --param 1.2 --batch 16384 --precision f32 --launches 2|sq hbm"
 "c4_stream_f32|--code gen:irg:10000 --alg MSA --channel biawgn --param 1.2 --batch 8192 --precision f32 --backend stream --launches 1 --warm 1|hbm"
 "c4_stream_f64|--code gen:irg:10000 --alg MSA --channel biawgn --param 1.2 --batch 8192 --precision f64 --backend stream --launches 1 --warm 1|hbm"
 "c5_stream_f32|--code gen:reg:64800:3:6 --alg MSA --channel biawgn --param 1.0 --batch 32768 --precision f32 --backend stream --launches 1 --warm 1|hbm"
 "c2_stream_f32|--code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f32 --backend stream --launches 1 --warm 1|hbm"
 "c2_stream_f64|--code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f64 --backend stream --launches 1 --warm 1|hbm"
 "c5_stream_f16|--code gen:reg:64800:3:6 --alg MSA --channel biawgn --param 1.0 --batch 8192 --precision f16 --launches 1 --warm 1|hbm"
 "c5_bec_stream|--code gen:reg:64800:3:6 --alg BEC --channel bec --param 0.40 --batch 32768 --precision f32 --launches 1 --warm 1|hbm"
 "c4_bec_stream|--code gen:irg:10000 --alg BEC --channel bec --param 0.44 --batch 32768 --precision f32 --launches 1 --warm 1|hbm"
)
for C in "${CASES[@]}"; do
  NAME=${C%%|*}; REST=${C#*|}; ARGS=${REST%%|*}; PASSES=${REST#*|}
  if [ $# -gt 0 ] && [[ ! " $* " =~ " $NAME " ]]; then continue; fi
  echo "== $NAME"
  mkdir -p $OUT/$NAME
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$NAME/stats -o k -- python3 $R/tools/sim_driver.py $ARGS --info $OUT/$NAME/stats.info.json > $OUT/$NAME.stats.log 2>&1
  if [[ $PASSES == *sq* ]]; then
    rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/$NAME/sq1 -o p -- python3 $R/tools/sim_driver.py $ARGS --info $OUT/$NAME/sq1.info.json > $OUT/$NAME.sq1.log 2>&1
    rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/$NAME/sq2 -o p -- python3 $R/tools/sim_driver.py $ARGS --info $OUT/$NAME/sq2.info.json > $OUT/$NAME.sq2.log 2>&1
    rocprofv3 --pmc $SQ4 --kernel-trace --output-format csv -d $OUT/$NAME/sq4 -o p -- python3 $R/tools/sim_driver.py $ARGS --info $OUT/$NAME/sq4.info.json > $OUT/$NAME.sq4.log 2>&1
    if [ -n "$SQ3" ]; then
      rocprofv3 --pmc $SQ3 --kernel-trace --output-format csv -d $OUT/$NAME/sq3 -o p -- python3 $R/tools/sim_driver.py $ARGS --info $OUT/$NAME/sq3.info.json > $OUT/$NAME.sq3.log 2>&1
    fi
  fi
  if [[ $PASSES == *hbm* ]]; then
    for CNT in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/$NAME/$CNT -o p -- python3 $R/tools/sim_driver.py $ARGS --calib --info $OUT/$NAME/$CNT.info.json > $OUT/$NAME.$CNT.log 2>&1
    done
  fi
  tail -1 $OUT/$NAME.stats.log | cut -c1-300
done
find $OUT -name "*.db" -delete
du -sh $OUT
