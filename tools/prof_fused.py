"""Small driver for profiler runs: N decodes of one resident batch (priors already in HBM), no CPU baseline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import argparse
import torch
from bench import load_code
from ldpc_decoders_amd._device import DecoderHandle

ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="auto"); ap.add_argument("--snr", type=float, default=1.0)
ap.add_argument("--batch", type=int, default=65536); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--code", default="1200_3_6_rand_ldpc_1"); ap.add_argument("--precision", default="f32")
ap.add_argument("--alg", default="MSA")
ap.add_argument("--info", default=None, help="write frames / sweeps of the profiled launches to this JSON file")
a = ap.parse_args()
code = load_code(a.code)
g = code
h = DecoderHandle(code, a.alg, a.precision, a.backend)
if a.alg == "BEC":
    pri, y = h.channel_device("bec", a.snr, 0, 1, 0, 0, a.batch)  # --snr carries the erasure probability
else:
    pri, y = h.channel_device("biawgn", a.snr, 0, 1, 0, 0, a.batch)
xh, it = h.decode_device(pri, y, 50)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    h.decode_device(pri, y, 50, xhat=xh, iters=it)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
# known-size 4 B/lane copy (calibration of FETCH_SIZE / WRITE_SIZE): 1 GiB in, 1 GiB out
from ldpc_decoders_amd import _lib
src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
_lib.check(_lib.load().ldpc_debug_copy4(src.data_ptr(), dst.data_ptr(), src.numel() * 4, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("backend", h.last_stats()[0], "ms/decode %.3f" % (dt * 1e3), "frames/s %.3e" % (a.batch / dt), "mean iters %.2f" % it.float().mean().item(), h.fused_info())
if a.info:
    import json
    # every launch (the warm-up one and the timed ones) decodes the same resident batch: frame-sweeps per launch = sum of iters
    json.dump({"batch": a.batch, "launches": a.reps + 1, "frame_sweeps_per_launch": int(it.sum().item()), "code": a.code, "snr": a.snr,
               "precision": a.precision, "alg": a.alg, "backend": h.last_stats()[0], "ms_per_decode_unprofiled_clock": dt * 1e3,
               "cus": torch.cuda.get_device_properties(0).multi_processor_count}, open(a.info, "w"))
