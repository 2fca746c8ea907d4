"""Throughput of the ML decoder path (channel -> exhaustive search -> count, all on the device) next to the numpy oracle
on one host core.  python tools/ml_rate.py [--frames N]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import argparse
import numpy as np
import torch
import ml_oracle as M
from ldpc_decoders_amd import codes
from ldpc_decoders_amd._device import MlHandle
from ldpc_decoders_amd.models import models

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=1 << 24); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
rows = []
for channel, code, param, prec in (("biawgn", "7_4_hamming", 2.0, "f32"), ("biawgn", "7_4_hamming", 2.0, "f64"), ("biawgn", "12_3_4_ldpc", 1.0, "f32"),
                                   ("bsc", "7_4_hamming", 0.1, "f32"), ("bec", "7_4_hamming", 0.3, "f32"), ("bsc", "12_3_4_ldpc", 0.1, "f32")):
    c = codes.get_code(code)
    h = MlHandle(c.cb, channel, prec)
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    h.simulate(channel, param, 0, 1, 0, 0, a.frames, 0, cnt)
    torch.cuda.synchronize(); cnt.zero_()
    t0 = time.perf_counter()
    for r in range(a.reps):
        h.simulate(channel, param, 0, 1, 0, r * a.frames, a.frames, 0, cnt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    tot, wec, bec = cnt.cpu().tolist()[:3]
    # numpy oracle, one core, bounded sample
    chan = models[channel].Channel(param)
    coef = M.ml_coefficients(channel, param)
    x = np.zeros(c.get_n(), dtype=np.int64)
    np.random.seed(1); nf = 0; t1 = time.perf_counter()
    with np.errstate(all="ignore"):
        while time.perf_counter() - t1 < 1.5:
            M.ml_decode(channel, c.cb, chan.send(x), coef); nf += 1
    cpu = nf / (time.perf_counter() - t1)
    rows.append(dict(channel=channel, code=code, param=param, obs=prec, K=int(c.cb.shape[0]), n=int(c.get_n()), frames=a.frames,
                     ms=dt * 1e3, frames_per_s=a.frames / dt, wer=wec / tot, ber=bec / (tot * c.get_n()), oracle_frames_per_s_1core=cpu))
    print(json.dumps(rows[-1]))
