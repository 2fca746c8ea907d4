#!/bin/bash
# Runs ON THE GPU BOX: everything a round's measurements are quoted from, under gpurun_out/<tag>/.   tools/round_measure.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1
python tools/measure_configs.py $TAG > $OUT/all_configs.log 2>&1; cp profiles/${TAG}_all_configs.json $OUT/ 2>/dev/null
( time python tools/compare_curves.py --out $OUT/curves_vs_reference.md ) > $OUT/curves.log 2>&1
( time python tools/compare_curves.py --precision f32 --out $OUT/curves_vs_reference_fp32.md ) > $OUT/curves_f32.log 2>&1
python tools/admm_rate.py > $OUT/admm_rate.log 2>&1
python tools/ml_rate.py > $OUT/ml_rate.log 2>&1
python tools/repack_probe.py 2.0 8192 > $OUT/repack_probe.log 2>&1
python tools/repack_probe.py 1.8 32768 gen:irg:10000 >> $OUT/repack_probe.log 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_admm -o a -- python3 $R/tools/admm_prof.py 8192 > $OUT/stats_admm.log 2>&1
find $OUT -name "*.db" -delete
tail -3 $OUT/gputest.log
