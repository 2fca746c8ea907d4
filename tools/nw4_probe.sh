#!/bin/bash
# Runs ON THE GPU BOX: the headline Monte-Carlo step (n = 1200 (3,6) min-sum, fp64, 1.0 dB, 65 536 frames) with two and with four waves per
# frame (LDPC_FUSED_NW picks the shape), three times each
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/tools/sim_driver.py --code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f64 --launches 6 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.4g frames/s  %.3f ms  %s' % (d['frames_per_s_wall'], d['ms_per_launch_wall'], d['kernel']), d.get('wer'), d.get('mean_sweeps'))"; }
for rep in 1 2 3; do
echo "NW=2: $(LDPC_FUSED_NW=2 run)"
echo "NW=4: $(LDPC_FUSED_NW=4 run)"
done
