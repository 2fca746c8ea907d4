#!/bin/bash
# Runs ON THE GPU BOX: the guarded (exact-in-fp32) Monte-Carlo kernels with library variants:  tools/ab_grid.sh TAG base NAME1 ...
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
export LDPC_FUSED_PLAN_DIR=$R/ldpc_decoders_amd/plans LDPC_FUSED_PLAN_SAVE=none
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
CASES=(
 "c4_grid|--code gen:irg:10000 --alg MSA --channel biawgn --param 1.2 --batch 32768 --precision f32 --prior-grid 8 --launches 2"
 "c4_plain|--code gen:irg:10000 --alg MSA --channel biawgn --param 1.2 --batch 32768 --precision f32 --launches 2"
 "c2_grid|--code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f32 --prior-grid 8 --launches 6"
 "c2_plain|--code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f32 --launches 6"
 "irr_grid|--code 1200_rho_x5_rand_ldpc_5 --alg MSA --channel biawgn --param 1.5 --batch 65536 --precision f32 --prior-grid 8 --launches 6"
)
for C in "${CASES[@]}"; do
  NAME=${C%%|*}; ARGS=${C#*|}
  for rep in 1 2; do for V in "$@"; do
    LIB=$R/ldpc_decoders_amd/csrc/variants/libldpc_hip_$V.so; [ $V = base ] && LIB=$R/ldpc_decoders_amd/csrc/libldpc_hip.so
    LDPC_LIB_PATH=$LIB python3 $R/tools/sim_driver.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$NAME $V %.4g frames/s  %.3f ms  %s' % (d['frames_per_s_wall'], d['ms_per_launch_wall'], d['kernel']))" >> $OUT/ab.txt
  done; done
done
cat $OUT/ab.txt
