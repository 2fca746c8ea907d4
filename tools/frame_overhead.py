"""Fixed per-frame cost of the fused kernels: time at exactly k sweeps (no early exit) for k = 1..16 -> slope & intercept, for the
decode entry (priors resident in HBM, decisions written) and for the in-kernel Monte-Carlo path (noise + decode + counting).
    python tools/frame_overhead.py [MSA|SPA] [f32|f64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_code
from ldpc_decoders_amd._device import DecoderHandle
code = load_code("1200_3_6_rand_ldpc_1")
alg = sys.argv[1] if len(sys.argv) > 1 else "MSA"
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
h = DecoderHandle(code, alg, prec, "auto")
B = 65536
pri, _ = h.channel_device("biawgn", 1.0, 0, 1, 0, 0, B)
xh, it = h.decode_device(pri, None, 2, flags=1)
cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
for name in ("decode", "simulate"):
    res = []
    for k in (1, 2, 4, 8, 16):
        def run():
            if name == "decode":
                h.decode_device(pri, None, k, flags=1, xhat=xh, iters=it)
            else:
                h.simulate("biawgn", 1.0, 0, 7, 0, 0, B, k, cnt, flags=1)
        run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize(); res.append((k, (time.perf_counter() - t0) / 5 * 1e3))
    (k1, t1), (k2, t2) = res[-2], res[-1]
    slope = (t2 - t1) / (k2 - k1)
    print("%s %s %s: %s  per sweep %.4f ms, per-frame fixed cost of 65536 frames %.4f ms = %.2f sweeps" % (
        alg, prec, name, [(k, round(t, 3)) for k, t in res], slope, res[0][1] - slope * res[0][0], (res[0][1] - slope) / slope))
