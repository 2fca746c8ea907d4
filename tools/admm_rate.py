"""Throughput of the ADMM LP decoder (device buffers in, estimates out) next to the C oracle on the host cores."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import admm_oracle as A
from ldpc_decoders_amd import codes
from ldpc_decoders_amd._device import AdmmHandle
os.environ.setdefault(codes.file_codes_dir_string, os.path.join(ROOT, "ldpc_decoders_amd", "data", "codes"))
for name, B, snr, mi in (("1200_3_6_rand_ldpc_1", 8192, 2.2, 300), ("1200_3_6_rand_ldpc_1", 65536, 2.2, 300), ("1200_3_6_rand_ldpc_1", 8192, 3.0, 300), ("7_4_hamming", 1 << 20, 2.0, 100)):
    code = codes.get_code(name)
    rng = np.random.RandomState(1)
    nv = 10 ** (-snr / 10)
    gamma = torch.from_numpy(-2 * (-1 + rng.normal(0, np.sqrt(nv), (B, code.n))) / nv).cuda()
    h = AdmmHandle(code)
    h.decode_device(gamma[:64], 3.0, 1e-5, mi)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, it, cv = h.decode_device(gamma, 3.0, 1e-5, mi)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    class G: m, n, chk, var = code.m, code.n, code.edge_chk, code.edge_var
    nb = min(B, 512)
    t1 = time.perf_counter(); A.admm_decode(G, gamma[:nb].cpu().numpy(), 3.0, 1e-5, mi); dc = time.perf_counter() - t1
    iters = it.float().mean().item()
    bytes_iter = 8 * (9 * code.E + 2 * code.n)  # x pass: z, lambda in, x out, gamma in; z pass: x gather, lambda in/out, z in/out, d1, d2 out; test: d1, d2 in
    print(json.dumps(dict(backend=h.last_backend(), code=name, frames=B, snr=snr, max_iter=mi, ms=dt * 1e3, frames_per_s=B / dt, mean_iters=iters, converged=cv.float().mean().item(),
                          frame_iters_per_s=B * (iters + 1) / dt, algorithmic_GBps=B * (iters + 1) * bytes_iter / dt / 1e9,
                          oracle_frames_per_s=nb / dc, oracle_threads=os.cpu_count())))
