#!/bin/bash
# Runs ON THE GPU BOX: fp64 sum-product Monte-Carlo step (n = 1200, 1.5 dB) with the built library and with variant libraries (names as arguments)
R=${GRAFT_REPO_ROOT:-/root/repo}
export LDPC_FUSED_PLAN_DIR=$R/ldpc_decoders_amd/plans LDPC_FUSED_PLAN_SAVE=none
run() { python3 $R/tools/sim_driver.py --code 1200_3_6_rand_ldpc_1 --alg SPA --channel biawgn --param 1.5 --batch 65536 --precision f64 --launches 2 "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.4g frames/s  %.3f ms  %s' % (d['frames_per_s_wall'], d['ms_per_launch_wall'], d['kernel']), 'wer', d.get('wer'), 'sweeps', d.get('mean_sweeps'))"; }
for rep in 1 2; do
echo "base fused : $(run)"
echo "base stream: $(run --backend stream)"
for V in "$@"; do
echo "$V fused : $(LDPC_LIB_PATH=$R/ldpc_decoders_amd/csrc/variants/libldpc_hip_$V.so run)"
echo "$V stream: $(LDPC_LIB_PATH=$R/ldpc_decoders_amd/csrc/variants/libldpc_hip_$V.so run --backend stream)"
done
done
