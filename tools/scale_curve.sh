#!/bin/bash
# The 1/2/4/8-GPU lines of bench.py, launched exactly as the round driver launches them (one rank per GPU, RCCL):
#   tools/scale_curve.sh [steps] [warmup] [extra bench.py flags ...]      -> one JSON line per N on stdout, gpurun_out/scale/N.json
# Weak scaling: every rank decodes 65 536 frames per step whatever N is; `value` is the whole-job rate.  Needs an N-GPU node.
STEPS=${1:-20}; WARMUP=${2:-5}; shift 2 2>/dev/null   # the round driver's own --steps 20 --warmup 5: the N = 1 line here == BENCH_rNN.json
R=$(cd "$(dirname "$0")/.." && pwd)
NG=$(python -c "import torch; print(torch.cuda.device_count())")
mkdir -p $R/gpurun_out/scale
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29700
for N in 1 2 4 8; do
  [ $N -gt $NG ] && { echo "{\"n_gpus\": $N, \"skipped\": \"node has $NG GPU(s)\"}"; continue; }
  if [ $N -eq 1 ]; then
    python $R/bench.py --gpus 1 --steps $STEPS --warmup $WARMUP "$@" | tee $R/gpurun_out/scale/$N.json
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((PORT + N)) $R/bench.py --gpus $N --steps $STEPS --warmup $WARMUP "$@" | grep '^{' | tee $R/gpurun_out/scale/$N.json
  fi
done
