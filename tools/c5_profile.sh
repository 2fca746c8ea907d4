cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05d
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05d/stats_c5 -o k -- python3 $R/bench.py --code gen:reg:64800:3:6 --batch 32768 --snr 2.0 --steps 2 --warmup 1 --repeats 1 --precision f32 --no-cpu-baseline --no-profile --points > $R/gpurun_out/r05d/stats_c5.log 2>&1
head -30 $R/gpurun_out/r05d/stats_c5/k_kernel_stats.csv
cd $R
python - <<'PY'
import torch, numpy as np, sys
sys.path.insert(0,'.')
from bench import load_code
from ldpc_decoders_amd._device import DecoderHandle
code=load_code("gen:reg:64800:3:6")
h=DecoderHandle(code,"MSA","f32","auto")
cnt=torch.zeros(4+60,dtype=torch.int64,device="cuda")
h.simulate("biawgn",2.0,0,0x5EED1200,0,0,32768,50,cnt,hist_bins=60)
c=cnt.cpu().numpy()
print("hist", c[4:].tolist(), "repacks", h.last_repacks())
PY
find $R/gpurun_out/r05d -name "*.db" -delete
