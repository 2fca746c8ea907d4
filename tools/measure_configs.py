"""One artifact for every BASELINE.json configuration that runs on a GPU (configs 2-5), single MI355X, device Monte-Carlo
(channel + decode + count on the GPU, `ldpc_simulate`).  Writes profiles/<tag>_all_configs.json.

    python tools/measure_configs.py [tag]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import HBM_PEAK_GBS, fused_roofline, load_code  # noqa: E402
from ldpc_decoders_amd._device import DecoderHandle  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
try:
    COUNTERS = json.load(open(os.path.join(ROOT, "profiles", "roofline_counters.json")))
except Exception:
    COUNTERS = {}
CUS = torch.cuda.get_device_properties(0).multi_processor_count
CASES = [  # config, code, decoder alg, channel, param, batch, steps, backend[, precision]
    ("2: n=1200 (3,6) MSA BI-AWGN, fp64 (the reference's arithmetic, bit-identical decisions)", "1200_3_6_rand_ldpc_1", "MSA", "biawgn", 1.0, 65536, 6, "auto", "f64"),
    ("2: n=1200 (3,6) MSA BI-AWGN, fp64 (the reference's arithmetic, bit-identical decisions)", "1200_3_6_rand_ldpc_1", "MSA", "biawgn", 2.0, 65536, 6, "auto", "f64"),
    ("2: n=1200 (3,6) MSA BI-AWGN, fp32 mode", "1200_3_6_rand_ldpc_1", "MSA", "biawgn", 1.0, 65536, 6, "auto"),
    ("2: n=1200 (3,6) MSA BI-AWGN, fp32 mode", "1200_3_6_rand_ldpc_1", "MSA", "biawgn", 2.0, 65536, 6, "auto"),
    ("2: same, exact-in-fp32 mode (priors on the 2^-8 grid, in-kernel guard, frames beyond it re-decoded in fp64: an fp64 run's counters)", "1200_3_6_rand_ldpc_1", "MSA", "biawgn", 1.0, 65536, 6, "auto", "f32", 8),
    ("2: same, exact-in-fp32 mode (priors on the 2^-8 grid, in-kernel guard, frames beyond it re-decoded in fp64: an fp64 run's counters)", "1200_3_6_rand_ldpc_1", "MSA", "biawgn", 2.0, 65536, 6, "auto", "f32", 8),
    ("3: n=1200 (3,6) SPA BSC", "1200_3_6_rand_ldpc_1", "SPA", "bsc", 0.07, 65536, 6, "auto"),
    ("3: n=1200 (3,6) SPA BSC", "1200_3_6_rand_ldpc_1", "SPA", "bsc", 0.05, 65536, 6, "auto"),
    ("3: n=1200 (3,6) SPA BI-AWGN, fp64 (the reference's formula verbatim, LDS kernel)", "1200_3_6_rand_ldpc_1", "SPA", "biawgn", 1.5, 65536, 3, "auto", "f64"),
    ("3: n=1200 (3,6) SPA BI-AWGN, fp64, streaming kernels", "1200_3_6_rand_ldpc_1", "SPA", "biawgn", 1.5, 16384, 1, "stream", "f64"),
    ("3: n=1200 (3,6) erasure decoder BEC, bit-sliced on the LDS, 2^20-frame launches (continuous refill wants long launches)", "1200_3_6_rand_ldpc_1", "BEC", "bec", 0.40, 1048576, 6, "auto"),
    ("3: n=1200 (3,6) erasure decoder BEC, bit-sliced on the LDS, 2^20-frame launches (continuous refill wants long launches)", "1200_3_6_rand_ldpc_1", "BEC", "bec", 0.35, 1048576, 6, "auto"),
    ("3: same, the BASELINE batch of 65 536 frames per launch (two slabs of 32 frames per workgroup: launch-bound)", "1200_3_6_rand_ldpc_1", "BEC", "bec", 0.40, 65536, 6, "auto"),
    ("3: same, bit-sliced streaming kernels (state in HBM)", "1200_3_6_rand_ldpc_1", "BEC", "bec", 0.40, 65536, 3, "stream"),
    ("4: rate-1/2 irregular n=10000 erasure decoder (bit-sliced streaming: a slab is 245 KB of bit planes)", "gen:irg:10000", "BEC", "bec", 0.44, 32768, 2, "auto"),
    ("5: (3,6) n=64800 erasure decoder (bit-sliced streaming)", "gen:reg:64800:3:6", "BEC", "bec", 0.40, 32768, 2, "auto"),
    ("4: rate-1/2 irregular n=10000 MSA (2^20 frames over 8 GPUs = 131072 per GPU)", "gen:irg:10000", "MSA", "biawgn", 1.2, 131072, 2, "auto"),
    ("4: rate-1/2 irregular n=10000 MSA (2^20 frames over 8 GPUs = 131072 per GPU)", "gen:irg:10000", "MSA", "biawgn", 1.8, 131072, 2, "auto"),
    ("4: same, exact-in-fp32 mode (2^-8 grid + guard + fp64 redo): the fp64 reference's counters from the LDS kernel", "gen:irg:10000", "MSA", "biawgn", 1.2, 131072, 2, "auto", "f32", 8),
    ("4: same, exact-in-fp32 mode (2^-8 grid + guard + fp64 redo): the fp64 reference's counters from the LDS kernel", "gen:irg:10000", "MSA", "biawgn", 1.8, 131072, 2, "auto", "f32", 8),
    ("4: same, streaming kernels", "gen:irg:10000", "MSA", "biawgn", 1.2, 32768, 1, "stream"),
    ("4: same, streaming kernels (frames leave after 11..50 sweeps: per-frame early termination by frame repack)", "gen:irg:10000", "MSA", "biawgn", 1.8, 32768, 1, "stream"),
    ("4: same, fp64 (the reference's arithmetic; 327 KB per frame: streaming kernels)", "gen:irg:10000", "MSA", "biawgn", 1.8, 32768, 1, "auto", "f64"),
    ("5: (3,6) n=64800 MSA, early termination (2^18 frames over 8 GPUs = 32768 per GPU)", "gen:reg:64800:3:6", "MSA", "biawgn", 1.0, 32768, 1, "auto"),
    ("5: (3,6) n=64800 MSA, early termination (2^18 frames over 8 GPUs = 32768 per GPU)", "gen:reg:64800:3:6", "MSA", "biawgn", 2.0, 32768, 1, "auto"),
    ("5: same, fp16 STORAGE mode (2-byte messages, fp32 arithmetic: tolerance mode, not the parity mode)", "gen:reg:64800:3:6", "MSA", "biawgn", 1.0, 32768, 1, "auto", "f16"),
    ("5: same, fp16 STORAGE mode (2-byte messages, fp32 arithmetic: tolerance mode, not the parity mode)", "gen:reg:64800:3:6", "MSA", "biawgn", 2.0, 32768, 1, "auto", "f16"),
    ("4: rate-1/2 irregular n=10000 MSA, fp16 STORAGE mode on the streaming kernels", "gen:irg:10000", "MSA", "biawgn", 1.8, 32768, 1, "stream", "f16"),
]
rows = []
cache = {}
for case in CASES:
    cfg, code_name, alg, ch, prm, B, steps, backend = case[:8]
    prec = case[8] if len(case) > 8 else "f32"
    grid = case[9] if len(case) > 9 else None
    redone = 0
    if code_name not in cache:
        cache[code_name] = load_code(code_name)
    code = cache[code_name]
    g = code
    h = DecoderHandle(code, alg, prec, backend)
    cnt = torch.zeros(4 + 51, dtype=torch.int64, device="cuda")
    def run(stream_id, frame0):
        global redone
        if grid is None:
            h.simulate(ch, prm, 0, 0x5EED1200, stream_id, frame0, B, 50, cnt, hist_bins=51)
        else:
            redone += h.simulate_exact_fp32(prm, 0, 0x5EED1200, stream_id, frame0, B, 50, cnt, grid, hist_bins=51)
    if grid is not None:
        h._fp64_sibling()  # the decoder that re-decodes the frames beyond the guard: built before the clock starts
    run(0, 0)  # warm-up at full size: workspaces are allocated here
    torch.cuda.synchronize()
    cnt.zero_()
    t0 = time.perf_counter()
    redone = 0
    for s in range(steps):
        run(1, s * B)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = cnt.cpu().numpy()
    frames, sweeps = int(c[0]), int(c[3])
    # algorithmic bytes per frame-sweep: SURVEY 8(d) s(4E + n); fp16 storage: 2-byte messages + 4-byte priors = 8E + 4n; erasure decoder:
    # 2 bits per message, (4E + m + 3n) / 4 bytes for the bit-sliced streaming sweep (ldpc_bec_stream.hip)
    bytes_fs = (8 if prec == "f64" else 4) * (4 * g.E + g.n)
    if prec == "f16":
        bytes_fs = 8 * g.E + 4 * g.n
    if alg == "BEC":
        bytes_fs = (4 * g.E + g.m + 3 * g.n) / 4.0
    # roofline of the dominant kernel: LDS-resident kernels from the committed PMC counters of that very kernel (keyed by its name) x the
    # frame-sweeps/s measured here; streaming kernels against the HBM peak with the section-8(d) algorithmic bytes of the sweep
    backend_used = h.last_stats()[0]
    if backend_used == "fused":
        kname = h.kernel_name(True)
        if grid is not None:
            kname = kname.replace("k_fused_bp<0, ", "k_fused_bp_grid<")
        roof = fused_roofline(kname, sweeps / dt, CUS, COUNTERS) or dict(bound="lds", frac=None, note="no committed counters for " + kname)
        roof = {k: roof.get(k) for k in ("bound", "frac", "lds_frac", "valu_frac", "lds_cycles_per_frame_sweep", "valu_busy_cycles_per_frame_sweep",
                                        "counters_workload", "counters_from", "note") if roof.get(k) is not None}
        roof["kernel"] = kname
        roof["note"] = "whole step (channel + decode + count are this one kernel): busy cycles per frame-sweep (PMC) x frame-sweeps/s of this run / available cycles at 2.4 GHz"
    else:
        gbs = sweeps * bytes_fs / dt / 1e9
        roof = dict(bound="hbm", frac=round(gbs / HBM_PEAK_GBS, 4), achieved_GBps=round(gbs, 1), peak_GBps=HBM_PEAK_GBS,
                    kernel="k_becs_cn + k_becs_vn" if alg == "BEC" else ("k_cn16 + k_vn16" if prec == "f16" else "k_cn + k_vn"),
                    counters_from="HBM bytes per launch of both passes: profiles/roofline_counters.json (hbm:* entries)",
                    note="whole step incl. channel, tile load, syndrome, repack and counting kernels: executed frame-sweeps x s(4E+n) / wall time / 8 TB/s")
    rows.append(dict(roofline=roof, config=cfg, code=code_name, n=g.n, E=g.E, decoder=alg, precision=prec, channel=ch, param=prm, max_iter=50, frames_per_step=B, steps=steps,
                     backend=h.last_stats()[0], repacks=h.last_repacks(), waves_per_frame=h.fused_info()["waves_per_frame"] if h.last_stats()[0] == "fused" else 0,
                     frames_per_s=round(frames / dt, 1), ms_per_step=round(1e3 * dt / steps, 3), mean_sweeps=round(sweeps / frames, 3),
                     wer=round(int(c[1]) / frames, 6), ber=int(c[2]) / (frames * g.n),
                     algorithmic_GBps=round(sweeps * bytes_fs / dt / 1e9, 1), **(dict(prior_grid="2^-%d" % grid, frames_redone_in_fp64=redone) if grid is not None else {})))
    print(json.dumps(rows[-1]), flush=True)
    del h
with open(os.path.join(ROOT, "profiles", "%s_all_configs.json" % tag), "w") as fp:
    json.dump(dict(device=torch.cuda.get_device_name(0), note="tools/measure_configs.py: whole hot path (device channel + decode + count), "
                   "max_iter 50; algorithmic_GBps = executed sweeps x s(4E+n) / time (SURVEY 8(d))", rows=rows), fp, indent=1)
