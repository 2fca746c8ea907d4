#!/usr/bin/env python3
"""Headline benchmark: decoded frames/s of the BP hot path on MI355X, n=1200 (3,6)-regular min-sum, max_iter=50.

A "step" is one pass of the hot path over one batch of synthetic input, entirely on the GPU:
    BI-AWGN channel + LLR kernel (Philox noise, all-zero word) -> flooding min-sum decode (syndrome early exit as in the
    reference) -> bit/word error counters.
Workload = BASELINE.json configs[1]: code 1200_3_6_rand_ldpc_1 (the reference's own H, tests/golden fixture), batch
65 536 frames per GPU.  Message arithmetic: fp64 by default -- the reference's own, hard decisions and iteration counts
bit-identical to it (LDS-resident fp64 min-sum kernel) --; `--precision f32` is the throughput mode, reported beside it under
"fp32_mode".  Default operating point 1.0 dB: every frame fails there, so every frame
executes exactly 50 sweeps -- the honest "50-iteration" number (no early-exit benefit).  `--snr` selects others;
`--points` adds 2.0/3.0 dB lines to the same JSON under "points".

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run, one rank per GPU
(RCCL); frames sharded by global frame index, ONE all-reduce of the counters per step; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def load_code(name):
    """Golden fixture name, or a generated ensemble member: 'gen:reg:<n>:<l>:<r>' / 'gen:irg:<n>' (BASELINE configs 4-5)."""
    from ldpc_decoders_amd import codes
    from ldpc_decoders_amd.codes import Code

    if name.startswith("gen:"):
        parts = name.split(":")
        rng = np.random.RandomState(20261002)
        if parts[1] == "reg":
            code = codes.rand_reg_ldpc(int(parts[2]), int(parts[3]), int(parts[4]), rng)
        else:
            code = codes.rand_irregular_ldpc(int(parts[2]), codes.LAMBDA_RHO_X5_HALF_RATE, 6, rng)

        class G:  # graph view for the C oracle
            m, n, E, chk, var = code.m, code.n, code.E, code.edge_chk, code.edge_var

        return G, code
    from helpers import golden_edges

    g = golden_edges(name)
    return g, Code.from_edges(g.m, g.n, g.chk, g.var)


def cpu_baseline(g, snr, max_iter, precision="f64", budget_s=12.0):
    """The CPU oracle (oracle/bp_oracle.c, a plain-C port of the reference algorithm, OpenMP over frames) timed on this
    host on a bounded sample of the same workload."""
    import bp_oracle as O
    import c_oracle as C

    cores = os.cpu_count() or 1
    dt_np = np.float64 if precision == "f64" else np.float32
    rng = np.random.RandomState(2024)
    var = O.biawgn_noise_var(snr)

    def sample(nf):
        y = -1 + rng.normal(0, np.sqrt(var), (nf, g.n))
        return O.biawgn_priors(y, snr).astype(dt_np)

    pri = sample(64 * cores)
    t0 = time.time()
    C.bp_decode(g, "MSA", None, pri, max_iter, dtype=dt_np, nthreads=cores)
    rate = len(pri) / max(time.time() - t0, 1e-6)
    nf = int(max(64 * cores, min(rate * budget_s, 400000)))
    pri = sample(nf)
    t0 = time.time()
    _, it = C.bp_decode(g, "MSA", None, pri, max_iter, dtype=dt_np, nthreads=cores)
    dt = time.time() - t0
    ref = {}
    try:
        with open(os.path.join(ROOT, "tests", "golden", "reference_timing.json")) as fp:
            tj = json.load(fp)
        for pt in tj["points"]:
            if pt["decoder"] == "MSA" and abs(pt["snr_db"] - snr) < 1e-9:
                ref = {"reference_python_frames_per_s_per_core": round(pt["frames_per_s"], 2),
                       "reference_python_measured_on": tj["host"]}
    except Exception:
        pass
    out = {"value": round(nf / dt, 1), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "%d frames, same H / SNR %.1f dB / max_iter %d, %s C port of the reference algorithm (oracle/bp_oracle.c), "
                     "%d OpenMP threads, %.1f s, mean %.1f sweeps/frame" % (nf, snr, max_iter, "fp64" if precision == "f64" else "fp32", cores, dt,
                                                                             float(it.mean()))}
    out.update(ref)
    return out


def run_point(sim, handle, comm, snr, steps, warmup, batch, stream_id, torch):
    """Times `steps` steps at one SNR; returns dict with time, counters and per-kernel event times."""
    per_round = batch * comm.world
    frame0 = 0
    for _ in range(warmup):
        sim.run_round(snr, stream_id, frame0, per_round)
        frame0 += per_round
    handle.read_profile(reset=True)
    tot = np.zeros(4 + sim.hist_bins, dtype=np.int64)
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tot += sim.run_round(snr, stream_id, frame0, per_round)
        frame0 += per_round
    torch.cuda.synchronize()
    comm.barrier()
    dt = comm.max_float(time.perf_counter() - t0)
    return dict(seconds=dt, counters=tot, profile=handle.read_profile(reset=True))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536, help="frames per GPU per step")
    ap.add_argument("--snr", type=float, default=1.0)
    ap.add_argument("--max-iter", type=int, default=50)
    ap.add_argument("--code", default="1200_3_6_rand_ldpc_1")
    ap.add_argument("--precision", default="f64", choices=["f32", "f64"],
                    help="message arithmetic; f64 is the reference's own (hard decisions bit-identical to it), f32 the throughput mode")
    ap.add_argument("--backend", default="auto", choices=["auto", "stream", "fused"])
    ap.add_argument("--points", type=float, nargs="*", default=[2.0, 3.0], help="extra SNR points reported under 'points'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event kernel timing (roofline leg)")
    args = ap.parse_args()

    import torch

    from ldpc_decoders_amd import dist
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.montecarlo import DeviceSimulator

    comm = dist.init_from_env()
    if comm.world != args.gpus and comm.is_root:
        print("warning: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, comm.world), file=sys.stderr)
    g, code = load_code(args.code)
    handle = DecoderHandle(code, "MSA", args.precision, args.backend)
    handle.set_profiling(not args.no_profile)
    sim = DeviceSimulator(handle, "biawgn", args.max_iter, 0, 0x5EED1200, comm, hist_bins=args.max_iter + 1)
    s = 8 if args.precision == "f64" else 4
    bytes_per_frame_iter = s * (4 * g.E + g.n)

    res = run_point(sim, handle, comm, args.snr, args.steps, args.warmup, args.batch, 0, torch)
    backend_used, _ = handle.last_stats()
    extra = []
    for i, snr in enumerate(args.points):
        r = run_point(sim, handle, comm, snr, max(2, args.steps // 2), 1, args.batch, 1 + i, torch)
        extra.append((snr, r))

    # The fused backend keeps the messages on-chip, so the HBM roofline does not bound it.  For an HBM-bound reading of the
    # same workload, time 2 steps of the streaming backend too (messages in HBM, [tile, edge, 64] layout) at N=1.
    stream_res = None
    if backend_used == "fused" and comm.world == 1 and not args.no_profile:
        h2 = DecoderHandle(code, "MSA", args.precision, "stream")
        h2.set_profiling(True)
        sim2 = DeviceSimulator(h2, "biawgn", args.max_iter, 0, 0x5EED1200, comm, hist_bins=args.max_iter + 1)
        stream_res = run_point(sim2, h2, comm, args.snr, 2, 1, args.batch, 0, torch)
        del sim2, h2

    # Same frames with the priors RESIDENT IN HBM when the timed region starts (channel kernel run beforehand): decode + count
    # only, rank 0's shard.  Reported beside `value` (which times the whole hot path: channel + decode + count).
    hbm_leg = None
    if comm.world == 1 and not args.no_profile:
        from ldpc_decoders_amd import _lib

        pri, _y = handle.channel_device("biawgn", args.snr, 0, 0x5EED1200, 0, 0, args.batch)
        xh, it = handle.decode_device(pri, None, args.max_iter)
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            handle.decode_device(pri, None, args.max_iter, xhat=xh, iters=it)
            _lib.check(_lib.load().ldpc_count_errors(xh.data_ptr(), None, 0, it.data_ptr(), args.batch, g.n, 0, cnt.data_ptr(), st))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        hbm_leg = {"frames_per_s": round(args.steps * args.batch / dt, 1), "ms_per_step": round(1e3 * dt / args.steps, 3),
                   "mean_sweeps": round(float(cnt[3]) / float(cnt[0]), 3),
                   "note": "priors [B,n] %s resident in HBM before the timed region; decode (ldpc_decode) + error counting (ldpc_count_errors)" % args.precision}
        del pri, xh, it

    # the fp32 throughput mode of the same workload (statistically identical curves, not bit-identical frame by frame), N = 1
    f32_res = None
    if args.precision == "f64" and comm.world == 1 and not args.no_profile:
        h3 = DecoderHandle(code, "MSA", "f32", args.backend)
        sim3 = DeviceSimulator(h3, "biawgn", args.max_iter, 0, 0x5EED1200, comm, hist_bins=args.max_iter + 1)
        f32_res = run_point(sim3, h3, comm, args.snr, args.steps, 1, args.batch, 0, torch)
        f32_res["backend"] = h3.last_stats()[0]
        del sim3, h3

    def summarise(snr, r, steps):
        c = r["counters"]
        frames, iter_sum = int(c[0]), int(c[3])
        fps = frames / r["seconds"]
        return {"snr_db": snr, "frames_per_s": round(fps, 1), "ms_per_step": round(1e3 * r["seconds"] / steps, 3),
                "mean_sweeps": round(iter_sum / max(frames, 1), 3), "wer": round(int(c[1]) / max(frames, 1), 6),
                "ber": float(c[2]) / max(frames * g.n, 1),
                "algorithmic_GBps": round(iter_sum * bytes_per_frame_iter / r["seconds"] / 1e9, 1)}

    if comm.is_root:
        head = summarise(args.snr, res, args.steps)
        c = res["counters"]
        iter_sum_rank0_share = int(c[3]) / comm.world  # profile is rank 0's; counters are whole-job
        prof = res["profile"]
        # dominant kernel = the class with the most event time on rank 0
        kind = max(prof, key=lambda k: prof[k][0])
        ms, launches = prof[kind]
        frac_bytes = {"stream_check_pass": 2 * g.E * s, "stream_variable_pass": (2 * g.E + g.n) * s,
                      "fused_decode": bytes_per_frame_iter}[kind]
        roof = None
        # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes, calibrated and
        # corrected as MI355X_MICROARCH.md prescribes): collected by tools/collect_profiles.sh, committed under profiles/
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fp:
                tj = json.load(fp)
            # keys of profiles/hbm_traffic.json: "<pass>:<precision>:<kernel>" (tools/summarize_profile.py)
            key = {"stream_check_pass": "stream:%s:k_cn" % args.precision, "stream_variable_pass": "stream:%s:k_vn" % args.precision,
                   "fused_decode": "sim:%s:%s" % (args.precision, "k_fused_msa64" if args.precision == "f64" else "k_fused_bp")}[kind]
            for k, v in tj.items():
                if k.startswith(key) and args.batch == 65536 and args.code == "1200_3_6_rand_ldpc_1" and abs(args.snr - 1.0) < 1e-9:
                    traffic = int(v)
        except Exception:
            pass
        if launches > 0 and ms > 0:
            bytes_total = iter_sum_rank0_share * frac_bytes
            ach = bytes_total / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": kind, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(bytes_total / launches), "avg_launch_ms": round(ms / launches, 4),
                    "launches": int(launches),
                    "note": "algorithmic bytes = sum over frames of sweeps executed x %d B (%s share of s(4E+n)); HIP events on the "
                            "decode stream, rank 0; traffic = PMC HBM bytes/launch from profiles/hbm_traffic.json (same workload). "
                            "frac > 1 means the messages stayed on-chip (fused backend): the kernel is then LDS/VALU-bound, "
                            "see DESIGN.md" % (frac_bytes, kind),
                    "all_kernels_ms": {k: round(v[0], 3) for k, v in prof.items()}}
        if roof is not None and kind == "fused_decode":
            # what actually bounds the on-chip kernel: LDS instruction issue.  Per frame-sweep the kernel issues the gathers (2 LDS
            # cycles each, conflict-free), their planned bank-conflict cycles, and one lane-contiguous store per message / marginal
            # row (2 cycles each; the 16-wave shape pairs its message rows at 3 cycles per row) -- all known from the plan.
            fi = handle.fused_info()
            nw, cr, vr_ = int(fi["waves_per_frame"]), int(fi["check_rounds"]), int(fi["variable_rounds"])
            row_cycles = 6 if args.precision == "f64" else 2  # ds_write_b64 rows vs ds_write_addtid_b32 rows
            store_cycles = (cr * 6 * (3 if nw > 4 else row_cycles)) + vr_ * row_cycles
            lds_cycles = fi["lds_gather_cycles_min"] + fi["conflict_cycles_planned"] + store_cycles
            frame_sweeps_per_s = iter_sum_rank0_share / (ms * 1e-3)
            cus, clk = torch.cuda.get_device_properties(0).multi_processor_count, 2.4e9
            roof["lds_pipe"] = {"bound": "lds", "unit": "LDS-array cycles/s per CU", "achieved": round(frame_sweeps_per_s * lds_cycles / cus, 1),
                                "peak": clk, "frac": round(frame_sweeps_per_s * lds_cycles / cus / clk, 4),
                                "cycles_per_frame_sweep": {"gathers": fi["lds_gather_cycles_min"], "planned_conflicts": fi["conflict_cycles_planned"],
                                                           "stores": store_cycles},
                                "note": "instruction-level model of the LDS pipe (MI355X_MICROARCH.md, LDS table) at a nominal 2.4 GHz; "
                                        "measured SQ_LDS_IDX_ACTIVE per frame-sweep is in profiles/*_summary.md"}
        out = {
            "metric": ("decoded frames/s, n=1200 (3,6) min-sum max_iter=50 (+ achieved HBM GB/s in roofline)"
                       if args.code == "1200_3_6_rand_ldpc_1" and args.max_iter == 50 else
                       "decoded frames/s, %s min-sum max_iter=%d (+ achieved HBM GB/s in roofline)" % (args.code, args.max_iter)),
            "value": head["frames_per_s"], "unit": "frames/s", "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%s MSA over BI-AWGN, max_iter=%d, batch=%d frames/GPU, %.1f dB (mean %.2f sweeps/frame), "
                                   "all-zero word + Philox noise on device" % (args.code, args.max_iter, args.batch, args.snr, head["mean_sweeps"]),
                       "code": args.code, "n": g.n, "m": g.m, "E": g.E, "decoder": "MSA", "channel": "biawgn", "snr_db": args.snr,
                       "max_iter": args.max_iter, "batch_per_gpu": args.batch, "backend": backend_used,
                       "parallelism": "frames sharded over %d GPU(s), 1 all-reduce of counters per step" % comm.world},
            "mean_sweeps": head["mean_sweeps"], "wer": head["wer"], "ber": head["ber"],
            "algorithmic_GBps": head["algorithmic_GBps"], "bytes_per_frame_sweep": bytes_per_frame_iter,
            "roofline": roof,
            "decode_from_hbm": hbm_leg,
            "fp32_mode": None if f32_res is None else dict(summarise(args.snr, f32_res, args.steps), backend=f32_res["backend"],
                                                           note="same workload with fp32 message arithmetic (bench.py --precision f32)"),
            "points": [summarise(snr, r, max(2, args.steps // 2)) for snr, r in extra],
        }
        if stream_res is not None:
            sp, sc = stream_res["profile"], stream_res["counters"]
            it_sum = int(sc[3])
            legs = {}
            for kname, share in (("stream_check_pass", 2 * g.E * s), ("stream_variable_pass", (2 * g.E + g.n) * s)):
                kms, kl = sp[kname]
                if kl:
                    gbs = it_sum * share / (kms * 1e-3) / 1e9
                    legs[kname] = {"achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "avg_launch_ms": round(kms / kl, 4),
                                   "launches": int(kl), "algorithmic_bytes_per_launch": int(it_sum * share / kl)}
            out["roofline_streaming_backend"] = {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernels": legs,
                "frames_per_s": round(int(sc[0]) / stream_res["seconds"], 1),
                "note": "same workload with --backend stream (messages resident in HBM): the HBM-bound path used for codes that do "
                        "not fit the LDS; PMC traffic per launch in profiles/hbm_traffic.json equals the algorithmic bytes"}
        if comm.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g, args.snr, args.max_iter, args.precision)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    dist.finalize()


if __name__ == "__main__":
    main()
