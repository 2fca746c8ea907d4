#!/bin/bash
# Runs ON THE GPU BOX: tools/ab_variants.sh TAG "prof_fused args" base NAME1 NAME2 ...  -- the same decode with each library variant
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; ARGS=$2; shift 2
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for V in "$@"; do
  LIB=$R/ldpc_decoders_amd/csrc/variants/libldpc_hip_$V.so
  [ $V = base ] && LIB=$R/ldpc_decoders_amd/csrc/libldpc_hip.so
  for rep in 1 2; do
    echo -n "$V " >> $OUT/ab.txt
    LDPC_LIB_PATH=$LIB python3 $R/tools/prof_fused.py $ARGS 2>/dev/null | tail -1 | cut -c1-90 >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
