R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/tools/sim_driver.py "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.4g frames/s  %.3f ms  %s' % (d['frames_per_s_wall'], d['ms_per_launch_wall'], d['kernel']), d.get('wer'), d.get('mean_sweeps'))"; }
for rep in 1 2; do for P in f32 f64; do for C in gen:reg:1200:4:8 gen:reg:1200:3:4; do
A="--code $C --alg MSA --channel biawgn --param 1.5 --batch 65536 --precision $P --launches 4"
echo "$C $P new : $(run $A)"
echo "$C $P head: $(LDPC_FUSED_PLAN_DIR=$R/ldpc_decoders_amd/plans LDPC_LIB_PATH=$R/ldpc_decoders_amd/csrc/variants/libldpc_hip_head.so run $A)"
done; done; done
