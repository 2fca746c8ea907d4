#!/usr/bin/env python3
"""One-frame latency of the registry path the reference's unchanged loop takes (src/main.py:37-48: one `decoder.decode(y)` per frame;
src/utils.py:84: the KAT harness): microseconds per call through `models[channel].<DEC>(param, code, **kwargs).decode(y)`, and of
the layers below it (ldpc_decode_host through ctypes; a device-resident ldpc_decode + synchronise).

    python tools/latency_probe.py [--code 1200_3_6_rand_ldpc_1] [--snr 2.0] [--calls 2000] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import load_code  # noqa: E402
from ldpc_decoders_amd.models import models  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--code", default="1200_3_6_rand_ldpc_1")
ap.add_argument("--snr", type=float, default=2.0)
ap.add_argument("--calls", type=int, default=2000)
ap.add_argument("--json", default=None)
a = ap.parse_args()
code = load_code(a.code)
x = np.zeros(code.n, dtype=np.int64)
rows = []
for dec_name, prec in (("MSA", "f64"), ("MSA", "f32"), ("SPA", "f32")):
    np.random.seed(1)
    mod = models["biawgn"]
    chan = mod.Channel(a.snr)
    dec = getattr(mod, dec_name)(a.snr, code, max_iter=50, precision=prec)
    ys = [chan.send(x) for _ in range(64)]
    for y in ys[:8]:
        dec.decode(y)  # warm-up: workspaces, plans
    torch.cuda.synchronize()
    t = []
    its = 0
    for i in range(a.calls):
        y = ys[i & 63]
        t0 = time.perf_counter()
        dec.decode(y)
        t.append(time.perf_counter() - t0)
        its += int(dec.dec.last_iters[0])
    t = np.array(t) * 1e6
    # the layer below: priors already formed, straight into the C ABI
    pri = [np.ascontiguousarray(dec.priors(y)[None, :], dtype=dec.dec.handle.np_dtype) for y in ys]
    h = dec.dec.handle
    t2 = []
    for i in range(a.calls):
        t0 = time.perf_counter()
        h.decode_host(pri[i & 63], None, 50)
        t2.append(time.perf_counter() - t0)
    t2 = np.array(t2) * 1e6
    # device-resident frame: ldpc_decode on the current stream + synchronise
    pd = [torch.from_numpy(p).cuda() for p in pri]
    xh = torch.empty((1, code.n), dtype=torch.uint8, device="cuda")
    it = torch.empty((1,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t3 = []
    for i in range(a.calls):
        t0 = time.perf_counter()
        h.decode_device(pd[i & 63], None, 50, xhat=xh, iters=it)
        torch.cuda.synchronize()
        t3.append(time.perf_counter() - t0)
    t3 = np.array(t3) * 1e6
    row = dict(decoder=dec_name, precision=prec, code=a.code, snr_db=a.snr, calls=a.calls, backend=h.last_stats()[0], mean_sweeps=its / a.calls,
               registry_decode_us=dict(median=round(float(np.median(t)), 1), p10=round(float(np.percentile(t, 10)), 1), p90=round(float(np.percentile(t, 90)), 1)),
               ldpc_decode_host_us=dict(median=round(float(np.median(t2)), 1), p10=round(float(np.percentile(t2, 10)), 1), p90=round(float(np.percentile(t2, 90)), 1)),
               device_resident_decode_plus_sync_us=dict(median=round(float(np.median(t3)), 1), p10=round(float(np.percentile(t3, 10)), 1), p90=round(float(np.percentile(t3, 90)), 1)))
    rows.append(row)
    print(json.dumps(row), flush=True)
if a.json:
    json.dump(dict(device=torch.cuda.get_device_name(0), note="tools/latency_probe.py: one frame per call, wall-clock microseconds on the host", rows=rows), open(a.json, "w"), indent=1)
