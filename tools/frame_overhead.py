"""Fixed per-frame cost of the fused kernel: decode time at exactly k sweeps (no early exit) for k = 1..4 -> slope & intercept."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from bench import load_code
from ldpc_decoders_amd._device import DecoderHandle
code = load_code("1200_3_6_rand_ldpc_1")
g = code
h = DecoderHandle(code, sys.argv[1] if len(sys.argv) > 1 else "MSA", "f32", "auto")
B = 65536
pri, _ = h.channel_device("biawgn", 1.0, 0, 1, 0, 0, B)
xh, it = h.decode_device(pri, None, 2, flags=1)
res = []
for k in (1, 2, 4, 8, 16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): h.decode_device(pri, None, k, flags=1, xhat=xh, iters=it)
    torch.cuda.synchronize(); res.append((k, (time.perf_counter() - t0) / 5 * 1e3))
print(res)
(k1, t1), (k2, t2) = res[-2], res[-1]
slope = (t2 - t1) / (k2 - k1)
print("per sweep %.4f ms, intercept (per-frame fixed cost of 65536 frames) %.4f ms = %.2f sweeps" % (slope, res[0][1] - slope * res[0][0], (res[0][1] - slope) / slope))
