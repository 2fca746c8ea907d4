#!/bin/bash
# Counterpart of the reference's run_sims.sh for the GPU build.
#   ./run_sims.sh SEQL REG_ENS --data_dir=./data --console     one run after another on GPU 0
#   ./run_sims.sh PARA REG_ENS --data_dir=./data               runs dealt round-robin over the node's GPUs, in parallel
# Code names resolve against $FILE_CODES_DIR, else ./data/codes, else the reference's data/codes files shipped inside the package.
MODE=${1:-SEQL}
CASE=$2
OTHER=${@:3:99}
NGPU=$(python -c "import torch; print(max(1, torch.cuda.device_count()))" 2>/dev/null || echo 1)
log () { echo "run|$CASE|$1"; }
i=0
while IFS= read -r line; do
    cmd="python -u -m ldpc_decoders_amd.main $line"
    if [ "$MODE" == "PARA" ]; then
        dev=$(( i % NGPU )); log ">> [gpu $dev] $cmd &"
        HIP_VISIBLE_DEVICES=$dev eval $cmd &
        i=$(( i + 1 )); if [ $(( i % NGPU )) -eq 0 ]; then wait; fi
    else
        log ">> $cmd"; eval $cmd
    fi
done < <( python -u -m ldpc_decoders_amd.simulations $CASE $OTHER )
log "Waiting..."; wait; log "Done!"
