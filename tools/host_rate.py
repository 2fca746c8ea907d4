"""PCIe-inclusive rate of the host-buffer entry point (ldpc_decode_host): numpy priors in, numpy decisions out."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from bench import load_code
from ldpc_decoders_amd import bpa
code = load_code("1200_3_6_rand_ldpc_1")
g = code
for snr in (1.0, 3.0):
    rng = np.random.RandomState(1)
    B = 65536
    pri = (-2 * (-1 + rng.normal(0, np.sqrt(10 ** (-snr / 10)), (B, g.n))) / 10 ** (-snr / 10)).astype(np.float32)
    dec = bpa.MSA(code, max_iter=50, precision="f32")
    dec.decode_batch(None, pri)
    t0 = time.perf_counter()
    for _ in range(3): x, it = dec.decode_batch(None, pri)
    dt = (time.perf_counter() - t0) / 3
    hb = dec.handle
    t0 = time.perf_counter()
    for _ in range(3): bits, _e, it2 = hb.decode_host_bits(pri, None, 50)
    dtb = (time.perf_counter() - t0) / 3
    assert (it2 == it).all()
    print("snr %.1f: ldpc_decode_host_bits (packed decisions stay packed) %.2f ms = %.2f M frames/s" % (snr, dtb * 1e3, B / dtb / 1e6))
    print("snr %.1f: host-buffer decode %.2f ms per %d frames = %.2f M frames/s (%.1f GB/s over PCIe, pageable numpy buffers), mean sweeps %.1f" % (snr, dt * 1e3, B, B / dt / 1e6, B * g.n * 5 / dt / 1e9, it.mean()))
