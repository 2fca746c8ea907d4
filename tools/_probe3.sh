R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03d
mkdir -p $OUT
cd $R
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[1], "FAILED", e); sys.exit()
r=d["roofline"]; k=r.get("all_kernels_ms")
print("%s: %.4g frames/s ms/step %.2f sweeps %.2f alg %.0f GB/s cn %.1f vn %.1f total %.1f" % (sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d["mean_sweeps"], d["algorithmic_GBps"], k['stream_check_pass'], k['stream_variable_pass'], k['stream_decode_total']))
PY
}
for CPW in 2 4 8; do for VPW in 8 16 32; do
  LDPC_STREAM_CPW=$CPW LDPC_STREAM_VPW=$VPW python bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 1.0 --steps 2 --warmup 1 --precision f32 --backend stream --no-cpu-baseline --points > $OUT/c5_cpw${CPW}_vpw${VPW}.json 2> $OUT/err.txt; summ $OUT/c5_cpw${CPW}_vpw${VPW}.json
done; done
for CPW in 2 4; do for VPW in 4 8 16; do
  LDPC_STREAM_CPW=$CPW LDPC_STREAM_VPW=$VPW python bench.py --batch 65536 --snr 1.0 --steps 3 --warmup 1 --precision f32 --backend stream --no-cpu-baseline --points > $OUT/c2_cpw${CPW}_vpw${VPW}.json 2> $OUT/err.txt; summ $OUT/c2_cpw${CPW}_vpw${VPW}.json
  LDPC_STREAM_CPW=$CPW LDPC_STREAM_VPW=$VPW python bench.py --code gen:irg:10000 --batch 32768 --snr 1.2 --steps 2 --warmup 1 --precision f32 --backend stream --no-cpu-baseline --points > $OUT/c4_cpw${CPW}_vpw${VPW}.json 2> $OUT/err.txt; summ $OUT/c4_cpw${CPW}_vpw${VPW}.json
done; done
LDPC_STREAM_CPW=4 LDPC_STREAM_VPW=8 python bench.py --batch 65536 --snr 1.0 --steps 3 --warmup 1 --precision f64 --backend stream --no-cpu-baseline --points > $OUT/c2_f64_cpw4.json 2> $OUT/err.txt; summ $OUT/c2_f64_cpw4.json
LDPC_STREAM_CPW=2 LDPC_STREAM_VPW=4 python bench.py --batch 65536 --snr 1.0 --steps 3 --warmup 1 --precision f64 --backend stream --no-cpu-baseline --points > $OUT/c2_f64_cpw2.json 2> $OUT/err.txt; summ $OUT/c2_f64_cpw2.json
