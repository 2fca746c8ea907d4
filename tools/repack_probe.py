"""Early-termination efficiency of the streaming backend on the (3,6) n = 64 800 shape: time, executed sweeps, histogram of
sweeps per frame and tile-sweeps actually streamed, with the frame repack on / off.   python tools/repack_probe.py [snr] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import load_code
from ldpc_decoders_amd._device import DecoderHandle

snr = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
code = load_code(sys.argv[3] if len(sys.argv) > 3 else "gen:reg:64800:3:6")
for mode in ("0", "1"):
    os.environ["LDPC_STREAM_REPACK"] = mode
    h = DecoderHandle(code, "MSA", "f32", "stream")
    pri, y = h.channel_device("biawgn", snr, 0, 0x5EED1200, 1, 0, B)
    xh, it = h.decode_device(pri, None, 50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        h.decode_device(pri, None, 50, xhat=xh, iters=it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    itc = it.cpu().numpy()
    tile_max = itc[: B // 64 * 64].reshape(-1, 64).max(axis=1)
    bytes_fs = 4 * (4 * code.E + code.n)
    print("repack=%s snr %.2f frames %d: %.1f ms, mean sweeps %.2f, sum over tiles of the tile maximum / 64 = %.2f (efficiency of unrepacked tiles %.3f), "
          "algorithmic %.0f GB/s, repacks %d" % (mode, snr, B, dt * 1e3, itc.mean(), tile_max.mean(), itc.mean() / tile_max.mean(),
                                                   itc.sum() * bytes_fs / dt / 1e9, h.last_repacks()))
    if mode == "0":
        print("  histogram of sweeps per frame:", dict(zip(*[a.tolist() for a in np.unique(itc, return_counts=True)])))
    del h, pri, xh, it
