#!/bin/bash
# Runs ON THE GPU BOX: tools/ab_lib.sh "<sim_driver args>" base NAME1 NAME2 ...  -- interleaved A/B of library variants (tools/build_variant.sh) on
# one configuration, three runs each; `base` = the shipped library
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; shift
export LDPC_FUSED_PLAN_DIR=$R/ldpc_decoders_amd/plans LDPC_FUSED_PLAN_SAVE=none
for rep in 1 2 3; do for V in "$@"; do
  LIB=$R/ldpc_decoders_amd/csrc/variants/libldpc_hip_$V.so; [ $V = base ] && LIB=$R/ldpc_decoders_amd/csrc/libldpc_hip.so
  LDPC_LIB_ALLOW_OLDER_ABI=1 LDPC_LIB_PATH=$LIB python3 $R/tools/sim_driver.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$V %.4g frames/s  %.3f ms  %s' % (d['frames_per_s_wall'], d['ms_per_launch_wall'], d['kernel']))"
done; done
