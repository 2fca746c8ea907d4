"""Ad-hoc GPU bring-up script (not a test): prints mismatch details instead of asserting."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bp_oracle as O, c_oracle as C
from helpers import golden_edges
from ldpc_decoders_amd import bpa, bec, bsc
from ldpc_decoders_amd.codes import Code

def code_of(name):
    g = golden_edges(name); return g, Code.from_edges(g.m, g.n, g.chk, g.var)

def cmp(tag, xh, it, xo, io):
    bad = (xh != xo).any(axis=1); ibad = it != io
    print("%-50s frames=%d xhat-mismatch=%d iters-mismatch=%d  mean-it dev %.2f ora %.2f" % (tag, len(it), bad.sum(), ibad.sum(), it.mean(), io.mean()), flush=True)
    if bad.any() or ibad.any():
        f = np.flatnonzero(bad | ibad)[:5]; print("   first bad frames", f, "dev it", it[f], "ora it", io[f], "nerr dev", (xh[f] != 0).sum(1), "ora", (xo[f] != 0).sum(1))

print(torch.cuda.get_device_name(0))
rng = np.random.RandomState(1)
for name, snr, B in (("7_4_hamming", 2.0, 500), ("512_3_6_rand_ldpc_2", 2.5, 130), ("1200_3_6_rand_ldpc_1", 2.0, 200), ("1200_rho_x5_rand_ldpc_5", 2.0, 100)):
    g, code = code_of(name)
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(snr)), (B, g.n)); pri = O.biawgn_priors(y, snr)
    for alg in ("MSA", "SPA"):
        for prec, dt in (("f64", np.float64), ("f32", np.float32)):
            for backend in ("stream", "auto"):
                try:
                    dec = getattr(bpa, alg)(code, max_iter=30, precision=prec, backend=backend)
                    t = time.time(); xh, it = dec.decode_batch(None, pri.astype(dt)); dt_s = time.time() - t
                    xo, io = C.bp_decode(g, alg, None, pri.astype(dt), 30, dtype=dt)
                    cmp("%s %s %s %s (%s, %.1f ms)" % (name, alg, prec, backend, dec.handle.last_stats()[0], dt_s * 1e3), xh, it, xo, io)
                except Exception as e:
                    print("EXC", name, alg, prec, backend, repr(e))
    yb = (rng.random_sample((B, g.n)) < 0.04).astype(np.int64); yb[:3] = 0
    for alg in ("MSA", "SPA"):
        dec = getattr(bsc, alg)(0.04, code, max_iter=30, precision="f64")
        xh, it = dec.decode_batch(yb); xo, io = C.bp_decode(g, alg, yb.astype(float), O.bsc_priors(yb, 0.04), 30)
        cmp("%s bsc %s f64" % (name, alg), xh, it, xo, io)
    ye = np.where(rng.random_sample((B, g.n)) < 0.4, 2, 0).astype(np.uint8)
    dec = bec.SPA(0.4, code, max_iter=30); xh, it = dec.decode_batch(ye); xo, io = C.bec_decode(g, ye, 30)
    cmp("%s bec" % name, xh, it, xo, io)
