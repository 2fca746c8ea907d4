#!/bin/bash
# Runs ON THE GPU BOX: the fp32 sum-product Monte-Carlo kernel (config 3, BSC p = 0.07) with library variants: tools/ab_spa.sh TAG base NAME1 ...
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
export LDPC_FUSED_PLAN_DIR=$R/ldpc_decoders_amd/plans LDPC_FUSED_PLAN_SAVE=none
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
ARGS="--code 1200_3_6_rand_ldpc_1 --alg SPA --channel bsc --param 0.07 --batch 65536 --precision f32 --launches 40"
for rep in 1 2; do for V in "$@"; do
  LIB=$R/ldpc_decoders_amd/csrc/variants/libldpc_hip_$V.so; [ $V = base ] && LIB=$R/ldpc_decoders_amd/csrc/libldpc_hip.so
  LDPC_LIB_PATH=$LIB python3 $R/tools/sim_driver.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('c3_spa $V %.4g frames/s  %.3f ms' % (d['frames_per_s_wall'], d['ms_per_launch_wall']))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt
