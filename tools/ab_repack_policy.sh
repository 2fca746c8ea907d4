#!/bin/bash
# Runs ON THE GPU BOX: repack policies of the streaming backend (runtime switch), configs 5 (2.0 dB) and 4 (fp32 / fp64, 1.8 dB)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
CASES=(
 "c5_2dB|--code gen:reg:64800:3:6 --alg MSA --channel biawgn --param 2.0 --batch 32768 --precision f32 --backend stream --launches 2"
 "c4_f64|--code gen:irg:10000 --alg MSA --channel biawgn --param 1.8 --batch 32768 --precision f64 --backend stream --launches 2"
 "c4_f32|--code gen:irg:10000 --alg MSA --channel biawgn --param 1.8 --batch 32768 --precision f32 --backend stream --launches 2"
 "c2_f32|--code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 2.0 --batch 65536 --precision f32 --backend stream --launches 3"
)
for C in "${CASES[@]}"; do NAME=${C%%|*}; ARGS=${C#*|}
  for rep in 1 2; do for POL in fill rent; do
    LDPC_STREAM_REPACK_POLICY=$POL python3 tools/sim_driver.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$NAME $POL %.4g frames/s  %.3f ms  repacks %d' % (d['frames_per_s_wall'], d['ms_per_launch_wall'], d['repacks_last']))"
  done; done
done
