#!/bin/bash
# Runs ON THE GPU BOX: tools/ab_sim.sh TAG base NAME1 NAME2 ...  -- the Monte-Carlo hot path (tools/sim_driver.py) of the fused configurations
# with each library variant (tools/build_variant.sh), two runs each
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
# a variant library lives in csrc/variants/: it would look for the shipped plans in csrc/plans (its own "../plans") and, not finding them,
# anneal a short plan of its own -- a handicap of several per cent that has nothing to do with the code under test
export LDPC_FUSED_PLAN_DIR=$R/ldpc_decoders_amd/plans
export LDPC_FUSED_PLAN_SAVE=none
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
CASES=(
 "c2_f64|--code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f64 --launches 6"
 "c5_stream|--code gen:reg:64800:3:6 --alg MSA --channel biawgn --param 1.0 --batch 16384 --precision f32 --backend stream --launches 1"
 "c4_stream|--code gen:irg:10000 --alg MSA --channel biawgn --param 1.8 --batch 32768 --precision f32 --backend stream --launches 1"
 "c4_f32|--code gen:irg:10000 --alg MSA --channel biawgn --param 1.2 --batch 32768 --precision f32 --launches 2"
 "c4_grid|--code gen:irg:10000 --alg MSA --channel biawgn --param 1.2 --batch 32768 --precision f32 --prior-grid 8 --launches 2"
 "c4_spa|--code gen:irg:10000 --alg SPA --channel biawgn --param 1.2 --batch 16384 --precision f32 --launches 2"
 "c3_bec|--code 1200_3_6_rand_ldpc_1 --alg BEC --channel bec --param 0.40 --batch 65536 --precision f32 --launches 6"
 "c3_spa_bsc|--code 1200_3_6_rand_ldpc_1 --alg SPA --channel bsc --param 0.07 --batch 65536 --precision f32 --launches 6"
 "c2_f32|--code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f32 --launches 6"
 "irr_msa|--code 1200_rho_x5_rand_ldpc_5 --alg MSA --channel biawgn --param 1.5 --batch 65536 --precision f32 --launches 6"
)
for C in "${CASES[@]}"; do
  NAME=${C%%|*}; ARGS=${C#*|}
  [ -n "$ONLY" ] && ! [[ $NAME =~ $ONLY ]] && continue   # ONLY='c2_f32|c4_f32' tools/ab_sim.sh ...
  for rep in 1 2 3; do
    for V in "$@"; do
    LIB=$R/ldpc_decoders_amd/csrc/variants/libldpc_hip_$V.so
    [ $V = base ] && LIB=$R/ldpc_decoders_amd/csrc/libldpc_hip.so
      LDPC_LIB_ALLOW_OLDER_ABI=1 LDPC_LIB_PATH=$LIB python3 $R/tools/sim_driver.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$NAME $V %.4g frames/s  %.3f ms  %s' % (d['frames_per_s_wall'], d['ms_per_launch_wall'], d['kernel']))" >> $OUT/ab.txt
    done
  done
done
cat $OUT/ab.txt
