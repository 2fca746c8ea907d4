R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log; tail -4 $OUT/gputest.log
python bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 30 --warmup 3 --precision f32 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err
python - <<'PY'
import json
for f in ("bench.json","bench_f32.json"):
    try:
        d=json.load(open("gpurun_out/r03e/"+f)); print(f, d["value"], d["ms_per_step"], [ (p["snr_db"], p["frames_per_s"]) for p in d["points"]], d.get("fp32_mode",{}) and d["fp32_mode"].get("frames_per_s"))
    except Exception as e: print(f, "FAILED", e)
PY
python tools/measure_configs.py r03e > $OUT/all_configs.log 2>&1; python - <<'PY'
import json
d=json.load(open("profiles/r03e_all_configs.json"))
for r in d["rows"]: print(r["config"][:60], r["precision"], r["param"], r["backend"], "%.4g frames/s" % r["frames_per_s"], r["mean_sweeps"], r["algorithmic_GBps"])
PY
cp profiles/r03e_all_configs.json $OUT/
