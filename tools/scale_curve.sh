#!/bin/bash
# The 1/2/4/8-GPU lines of bench.py, launched exactly as the round driver launches them (one rank per GPU, RCCL over xGMI):
#   tools/scale_curve.sh [--config 2|3spa|3bec|4|5] [steps] [warmup] [extra bench.py flags ...]
#       -> one JSON line per N on stdout, gpurun_out/scale_<config>/N.json
# --config 2 (default): weak scaling of the headline -- every rank decodes 65 536 frames per step whatever N is (the round driver's own
#     SCALE run; its N = 1 line == BENCH_rNN.json).
# --config 4 / 5: the workloads BASELINE.json states for the WHOLE 8-GPU node -- 2^20 frames of the n = 10 000 irregular code, 2^18 frames
#     of the n = 64 800 (3,6) code with early termination -- as STRONG scaling: bench.py --total-batch T keeps the total per step fixed
#     and every rank decodes its shard (Comm.shard), `"scaling": "strong"` on the line.  (n = 64 800 fp32 needs 1.3 MB of state per frame:
#     one GPU holds ~98 304 frames at a time and takes 2^18 in three passes, ldpc_api.hip stream_chunk_frames.)
# --config 3spa / 3bec: config 3's two decoders, weak scaling at 65 536 frames per GPU.
# Needs an N-GPU node; Ns beyond the node's GPUs are reported as skipped.  value = whole-job frames/s; the driver computes efficiency.
CONFIG=2
if [ "$1" = "--config" ]; then CONFIG=$2; shift 2; fi
STEPS=${1:-}; WARMUP=${2:-}; shift 2 2>/dev/null
case $CONFIG in
  2)    FLAGS=""; S=20; W=5 ;;
  3spa) FLAGS="--decoder SPA --channel bsc --param 0.07 --precision f32"; S=20; W=3 ;;
  3bec) FLAGS="--decoder SPA --channel bec --param 0.40"; S=64; W=8 ;;
  4)    FLAGS="--code gen:irg:10000 --total-batch 1048576 --snr 1.2 --precision f32 --points 1.8 --repeats 3"; S=3; W=1 ;;
  5)    FLAGS="--code gen:reg:64800:3:6 --total-batch 262144 --snr 2.0 --precision f32 --points 1.0 --repeats 3"; S=2; W=1 ;;
  *) echo "unknown --config $CONFIG" >&2; exit 2 ;;
esac
STEPS=${STEPS:-$S}; WARMUP=${WARMUP:-$W}
R=$(cd "$(dirname "$0")/.." && pwd)
NG=$(python -c "import torch; print(torch.cuda.device_count())")
OUT=$R/gpurun_out/scale_$CONFIG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29700
for N in 1 2 4 8; do
  [ $N -gt $NG ] && { echo "{\"n_gpus\": $N, \"config\": \"$CONFIG\", \"skipped\": \"node has $NG GPU(s)\"}"; continue; }
  if [ $N -eq 1 ]; then
    python $R/bench.py --gpus 1 --steps $STEPS --warmup $WARMUP $FLAGS "$@" | tee $OUT/$N.json
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((PORT + N)) $R/bench.py --gpus $N --steps $STEPS --warmup $WARMUP $FLAGS "$@" | grep '^{' | tee $OUT/$N.json
  fi
done
