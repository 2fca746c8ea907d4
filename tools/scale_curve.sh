#!/bin/bash
# The 1/2/4/8-GPU lines of bench.py: tools/scale_curve.sh [--config 2|3spa|3bec|4|5] [steps] [warmup] [extra bench.py flags ...]
#       -> one JSON line per N on stdout, gpurun_out/scale_<config>/N.json
# bench.py starts its own ranks (`python bench.py --gpus N`: self_launch -> torch.distributed.run child, one rank per GPU, RCCL over xGMI),
# so every N is the plain command; a node with fewer than N GPUs ends that N with exit code 3 and no line.
# --config 2 (default): weak scaling of the headline, 65 536 frames per rank and step (the round driver's own SCALE run); its line also
#     carries `baseline_configs` = configs 3-5 at the same N (weak: 131 072 / 32 768 frames per rank = BASELINE's 2^20 / 2^18 at N = 8).
# --config 4 / 5: the same two workloads as STRONG scaling (--total-batch 2^20 / 2^18 whatever N is).  --config 3spa / 3bec: weak.
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
OUT=$R/gpurun_out/scale_$CONFIG
mkdir -p $OUT
for N in 1 2 4 8; do
  python $R/bench.py --gpus $N --steps $STEPS --warmup $WARMUP $FLAGS "$@" | tee $OUT/$N.json
  [ ${PIPESTATUS[0]} -eq 3 ] && echo "{\"n_gpus\": $N, \"config\": \"$CONFIG\", \"skipped\": \"bench.py exit 3: fewer than $N GPUs\"}"
done
