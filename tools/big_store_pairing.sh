#!/bin/bash
# A/B on the GPU box (VERDICT r3 task 9): the 16-wave n = 10 000 kernel with its paired row stores (ds_write2st64_b32, shipped) against
# two ds_write_b32 per pair (variant library built by tools/build_variant.sh bigunpaired "-DLDPC_BIG_UNPAIRED_STORES" ldpc_fused_shapes_f32_dc6.hip):
# time per launch and SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per frame-sweep, same box, same plan.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/bigpair; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--code gen:irg:10000 --alg MSA --channel biawgn --param 1.2 --batch 16384 --precision f32 --launches 3"
for V in shipped bigunpaired; do
  [ $V = shipped ] && unset LDPC_LIB_PATH || export LDPC_LIB_PATH=$R/ldpc_decoders_amd/csrc/variants/libldpc_hip_$V.so
  python3 $R/tools/sim_driver.py $ARGS --info $OUT/$V.info.json | tail -1 | cut -c1-200
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --kernel-trace --output-format csv -d $OUT/$V -o p -- python3 $R/tools/sim_driver.py $ARGS --info $OUT/$V.pmc.info.json > $OUT/$V.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
for v in ("shipped", "bigunpaired"):
    inf = json.load(open("$OUT/%s.info.json" % v)); pi = json.load(open("$OUT/%s.pmc.info.json" % v))
    acc = collections.defaultdict(float)
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if "k_fused_bp" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
    fs = float(pi["frame_sweeps"])
    print("%-12s %.3f ms per launch (un-instrumented), %.2f M frames/s | per frame-sweep: LDS-array cycles %.1f, bank-conflict cycles %.1f, DS instructions %.1f, data-FIFO-full %.1f, cmd-FIFO-full %.1f" % (
        v, inf["ms_per_launch_wall"], inf["frames_per_s_wall"] / 1e6, acc["SQ_LDS_IDX_ACTIVE"] / fs, acc["SQ_LDS_BANK_CONFLICT"] / fs, acc["SQ_INSTS_LDS"] / fs,
        acc["SQ_LDS_DATA_FIFO_FULL"] / fs, acc["SQ_LDS_CMD_FIFO_FULL"] / fs))
PY
