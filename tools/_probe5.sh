R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "SQ_INSTS_VALU|SQ_ACTIVE_INST|SQ_VALU|TRANS|F64|SQ_INST_CYCLES|SQ_BUSY" | cut -c1-160 | head -60 > $OUT/counters.txt
for X in 0 1; do
 for C in "c2|--code 1200_3_6_rand_ldpc_1 --batch 65536 --param 1.0" "c4|--code gen:irg:10000 --batch 16384 --param 1.2" "c5|--code gen:reg:64800:3:6 --batch 8192 --param 1.0"; do
  NAME=${C%%|*}; ARGS=${C#*|}
  for rep in 1 2; do
   LDPC_STREAM_XCD=$X python3 $R/tools/sim_driver.py $ARGS --alg MSA --channel biawgn --precision f32 --backend stream --launches 2 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$NAME xcd=$X %.4g frames/s  %.3f ms' % (d['frames_per_s_wall'], d['ms_per_launch_wall']))" >> $OUT/xcd.txt
  done
  mkdir -p $OUT/${NAME}_x$X
  LDPC_STREAM_XCD=$X rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${NAME}_x$X/FETCH_SIZE -o p -- python3 $R/tools/sim_driver.py $ARGS --alg MSA --channel biawgn --precision f32 --backend stream --launches 1 --calib > $OUT/${NAME}_x$X.log 2>&1
 done
done
cat $OUT/xcd.txt
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for d in sorted(glob.glob(R+"/gpurun_out/r03i/*_x*/FETCH_SIZE")):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
    print(d.split("/")[-2], {k: round(sum(v)/len(v)*2048/1e9,3) for k,v in acc.items() if "k_cn" in k or "k_vn" in k or "copy4" in k}, "GB fetched per launch")
PY
find $OUT -name "*.db" -delete
