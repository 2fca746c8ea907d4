#!/usr/bin/env python3
"""Profiler driver: N launches of the whole hot path (`ldpc_simulate`: device channel + decode + count) of one configuration -- the
very kernels bench.py / main.py time -- plus the bookkeeping a counter summary needs (frame-sweeps of the profiled launches, kernel
name).  Run under `rocprofv3 --pmc ...` by tools/collect_rooflines.sh.

    python3 tools/sim_driver.py --code 1200_3_6_rand_ldpc_1 --alg MSA --channel biawgn --param 1.0 --batch 65536 --precision f64 \
            --launches 3 --info out.json [--calib]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import load_code  # noqa: E402
from ldpc_decoders_amd import _lib  # noqa: E402
from ldpc_decoders_amd._device import DecoderHandle  # noqa: E402
from bench import AuxHandle  # noqa: E402  (ADMM / ML: the adapter bench.py times)

ap = argparse.ArgumentParser()
ap.add_argument("--code", default="1200_3_6_rand_ldpc_1")
ap.add_argument("--alg", default="MSA")
ap.add_argument("--channel", default="biawgn")
ap.add_argument("--param", type=float, default=1.0)
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--precision", default="f32")
ap.add_argument("--backend", default="auto")
ap.add_argument("--max-iter", type=int, default=50)
ap.add_argument("--launches", type=int, default=3)
ap.add_argument("--warm", type=int, default=6, help="launches BEFORE the counted ones (clocks up, workspaces allocated); the summary uses the last --launches dispatches only")
ap.add_argument("--info", default=None)
ap.add_argument("--prior-grid", type=int, default=None, help="exact-in-fp32 mode: priors on the 2^-K grid (LDPC_FLAG_PRIOR_GRID): the guarded fp32 kernel")
ap.add_argument("--calib", action="store_true", help="also run the known 1 GiB -> 1 GiB copy (calibration of FETCH_SIZE / WRITE_SIZE)")
a = ap.parse_args()

code = load_code(a.code)
aux = a.alg in ("ADMM", "ML")
h = AuxHandle(code, a.alg, a.precision, a.channel) if aux else DecoderHandle(code, a.alg, a.precision, a.backend)
bins = min(a.max_iter + 1, 60) if aux else a.max_iter + 1
flags = _lib.flag_prior_grid(a.prior_grid) if a.prior_grid is not None else 0
cnt = torch.zeros(4 + bins, dtype=torch.int64, device="cuda")
for w in range(max(1, a.warm)):  # warm-up: workspaces are allocated by the first launch, the clocks come up over the next ones
    h.simulate(a.channel, a.param, 0, 0x5EED1200, 0, w * a.batch, a.batch, a.max_iter, cnt, flags=flags, hist_bins=bins)
torch.cuda.synchronize()
cnt.zero_()  # frames / frame-sweeps below are those of the COUNTED launches (tools/summarize_rooflines.py takes the last `launches` dispatches)
t0 = time.perf_counter()
for s in range(a.launches):
    h.simulate(a.channel, a.param, 0, 0x5EED1200, 1, s * a.batch, a.batch, a.max_iter, cnt, flags=flags, hist_bins=bins)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / max(a.launches, 1)
if a.calib:
    src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    _lib.check(_lib.load().ldpc_debug_copy4(src.data_ptr(), dst.data_ptr(), src.numel() * 4, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
c = cnt.cpu().numpy()
backend = h.last_stats()[0]
kname = (h.kernel_name(True) if hasattr(_lib.load(), "ldpc_decoder_kernel_name") else "?") if backend == "fused" else (h.kernel_name(True) if aux else "")
if a.prior_grid is not None:  # the guarded sibling of that kernel (min-sum only: one template argument less)
    kname = kname.replace("k_fused_bp<0, ", "k_fused_bp_grid<")
info = dict(code=a.code, n=code.n, m=code.m, E=code.E, alg=a.alg, channel=a.channel, param=a.param, batch=a.batch, precision=a.precision,
            max_iter=a.max_iter, backend=backend, kernel=kname, launches=a.launches, warm_launches=max(1, a.warm),
            frames=int(c[0]), mean_sweeps=float(c[3]) / max(int(c[0]), 1), wer=float(c[1]) / max(int(c[0]), 1),
            # unit of work the counters are divided by: frame-sweeps (BP), frame-ITERATIONS incl. the one that meets the test (ADMM), frames (ML)
            frame_sweeps=int(c[3]) + int(c[0]) if a.alg == "ADMM" else (int(c[0]) if a.alg == "ML" else int(c[3])),
            ms_per_launch_wall=dt * 1e3, frames_per_s_wall=a.batch / dt, cus=torch.cuda.get_device_properties(0).multi_processor_count,
            repacks_last=0 if aux else h.last_repacks(), device=torch.cuda.get_device_name(0))
print(json.dumps(info))
if a.info:
    json.dump(info, open(a.info, "w"))
