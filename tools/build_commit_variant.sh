#!/bin/bash
# tools/build_commit_variant.sh NAME COMMIT -- builds ldpc_decoders_amd/csrc/variants/libldpc_hip_NAME.so from the library sources of COMMIT
# (for same-box A/B runs of the history: tools/ab_sim.sh; the variant finds the shipped plans through LDPC_FUSED_PLAN_DIR, which ab_sim.sh sets)
set -e
NAME=$1; COMMIT=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/hist_XXXX)
git -C $ROOT archive $COMMIT ldpc_decoders_amd/csrc include | tar -x -C $T
make -s -C $T/ldpc_decoders_amd/csrc -j${JOBS:-8} libldpc_hip.so
mkdir -p $ROOT/ldpc_decoders_amd/csrc/variants
cp $T/ldpc_decoders_amd/csrc/libldpc_hip.so $ROOT/ldpc_decoders_amd/csrc/variants/libldpc_hip_$NAME.so
rm -rf $T
echo $ROOT/ldpc_decoders_amd/csrc/variants/libldpc_hip_$NAME.so
