#!/bin/bash
# config 5 (n = 64 800) quick check ON THE GPU BOX: parity tests of the streaming backend, then the 2.0 dB / 1.0 dB bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r05e}; mkdir -p $R/gpurun_out/$TAG; cd $R
timeout 900 python -m pytest tests/test_gpu_repack.py tests/test_gpu_large_codes.py tests/test_gpu_f16_storage.py tests/test_gpu_parity.py tests/test_gpu_packed_bits.py tests/test_gpu_channel_sim.py -m gpu -x -q > gpurun_out/$TAG/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/gputest.log
timeout 600 python bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 2.0 --steps 2 --warmup 1 --repeats 3 --precision f32 --points 1.0 --no-cpu-baseline > gpurun_out/$TAG/bench_c5.json 2> gpurun_out/$TAG/bench_c5.err
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench_c5.json")); r=d["roofline"]
print("c5 2dB", d["value"], d["ms_per_step"], "kernel", d["kernel_ms_per_step"], "side", d["side_kernels_ms_per_step"], "host", d["host_overhead_ms_per_step"], "frac", r["frac"], {k:(v["frac"],v["avg_launch_ms"]) for k,v in r["passes"].items()})
print("whole-step frac", d["algorithmic_GBps"]/8000)
print("points", [(p["snr_db"],p["frames_per_s"],p["algorithmic_GBps"]/8000) for p in d["points"]])
f=d.get("fp16_storage_mode"); print("f16", f and (f["frames_per_s"], f["ms_per_step"], f["sweep_frac_of_hbm_peak"]))
PY
