#!/usr/bin/env python3
"""Headline benchmark: decoded frames/s of the BP hot path on MI355X, n=1200 (3,6)-regular min-sum, max_iter=50.

A "step" is one pass of the hot path over one batch of synthetic input, entirely on the GPU:
    BI-AWGN channel + LLR (Philox noise, all-zero word) -> flooding min-sum decode (syndrome early exit as in the
    reference) -> bit/word error counters.
Workload = BASELINE.json configs[1]: code 1200_3_6_rand_ldpc_1 (the reference's own H, shipped in ldpc_decoders_amd/data/codes), batch
65 536 frames per GPU.  Message arithmetic: fp64 by default -- the reference's own, hard decisions and iteration counts
bit-identical to it (LDS-resident fp64 min-sum kernel) --; `--precision f32` is the throughput mode, reported beside it under
"fp32_mode".  Default operating point 1.0 dB: every frame fails there, so every frame executes exactly 50 sweeps -- the honest
"50-iteration" number (no early-exit benefit).  `--snr` selects others; `--points` adds 2.0/3.0 dB lines under "points".

Timing.  `value` comes from a region with NO per-kernel instrumentation: K steps enqueued through the pipelined driver
(montecarlo.DeviceSimulator: two rounds in flight, counters all-reduced on the stream, no host sync between kernels), bracketed
by barrier + torch.cuda.synchronize() on both sides, MAX over ranks.  The per-kernel durations behind `roofline` are measured
in a SEPARATE pass of the same steps with the library's HIP events on the decode stream (`ldpc_decoder_profile`).

Roofline.  `roofline.bound` names the resource that binds the dominant kernel: "lds" for the LDS-resident (fused) kernels --
frac = LDS-array busy cycles / available cycles, numerator from the committed SQ_LDS_IDX_ACTIVE counters (profiles/lds_cycles.json,
written by tools/summarize_profile.py) x the frame-sweeps/s measured live -- and "hbm" for the streaming kernels (algorithmic
bytes of SURVEY.md 8(d) / HIP-event time / 8 TB/s).  The 8(d) HBM-model figure of the fused kernel is kept as `hbm_model`
(flagged: the messages never leave the CU, so it exceeds the HBM peak and bounds nothing).

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run, one rank per GPU
(RCCL); frames sharded by global frame index, ONE all-reduce of the counters per step; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
NOMINAL_CLOCK_HZ = 2.4e9  # max shader clock; the effective clock of a profiled run is in profiles/lds_cycles.json
KERNEL_CLASSES = ("stream_check_pass", "stream_variable_pass", "fused_decode")  # per-kernel HIP-event classes of ldpc_decoder_profile


def load_code(name):
    """A code file of the reference (its data/codes files ship inside the package), or a generated ensemble member:
    'gen:reg:<n>:<l>:<r>' / 'gen:irg:<n>' (BASELINE configs 4-5)."""
    from ldpc_decoders_amd import codes

    if name.startswith("gen:"):
        parts = name.split(":")
        rng = np.random.RandomState(20261002)
        if parts[1] == "reg":
            return codes.rand_reg_ldpc(int(parts[2]), int(parts[3]), int(parts[4]), rng)
        return codes.rand_irregular_ldpc(int(parts[2]), codes.LAMBDA_RHO_X5_HALF_RATE, 6, rng)
    return codes.load_parity_mtx(os.path.join(codes.PACKAGE_CODES_DIR, name + ".txt"))


def cpu_baseline(code, snr, max_iter, precision="f64", budget_s=12.0):
    """The CPU oracle (oracle/bp_oracle.c, a plain-C port of the reference algorithm, OpenMP over frames) timed on this
    host on a bounded sample of the same workload.  The only leg of this file that touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bp_oracle as O
    import c_oracle as C

    g = O.Edges(code.m, code.n, code.edge_chk, code.edge_var)
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


def run_point(sim, handle, comm, snr, steps, warmup, batch, stream_id, torch, kernel_pass=True):
    """Times `steps` steps at one SNR (uninstrumented, pipelined), then -- separately -- repeats them with the library's
    HIP-event kernel timing switched on.  Returns dict(seconds, counters, profile)."""
    per_round = batch * comm.world
    frame0 = 0
    handle.set_profiling(False)
    for _ in range(warmup):
        sim.run_round(snr, stream_id, frame0, per_round)
        frame0 += per_round
    tot = np.zeros(4 + sim.hist_bins, dtype=np.int64)
    first = frame0
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inflight = []
    for _ in range(steps):
        if len(inflight) == sim.DEPTH:
            tot += sim.finish_round(inflight.pop(0))
        inflight.append(sim.launch_round(snr, stream_id, frame0, per_round))
        frame0 += per_round
    while inflight:
        tot += sim.finish_round(inflight.pop(0))
    torch.cuda.synchronize()
    comm.barrier()
    dt = comm.max_float(time.perf_counter() - t0)
    prof = None
    if kernel_pass:  # same frames again, one step at a time, with HIP events around the dominant kernels (rank-local)
        handle.set_profiling(True)
        handle.read_profile(reset=True)
        f = first
        for _ in range(steps):
            sim.run_round(snr, stream_id, f, per_round)
            f += per_round
        prof = handle.read_profile(reset=True)
        handle.set_profiling(False)
    return dict(seconds=dt, counters=tot, profile=prof)


def committed(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fp:
            return json.load(fp)
    except Exception:
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="frames per GPU per step")
    ap.add_argument("--snr", type=float, default=1.0)
    ap.add_argument("--max-iter", type=int, default=50)
    ap.add_argument("--code", default="1200_3_6_rand_ldpc_1")
    ap.add_argument("--precision", default="f64", choices=["f32", "f64"],
                    help="message arithmetic; f64 is the reference's own (hard decisions bit-identical to it), f32 the throughput mode")
    ap.add_argument("--backend", default="auto", choices=["auto", "stream", "fused"])
    ap.add_argument("--points", type=float, nargs="*", default=[2.0, 3.0], help="extra SNR points reported under 'points'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="headline only: skip the HIP-event kernel pass and the side legs")
    args = ap.parse_args()

    import torch

    from ldpc_decoders_amd import dist
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.montecarlo import DeviceSimulator

    comm = dist.init_from_env()
    if comm.world != args.gpus and comm.is_root:
        print("warning: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, comm.world), file=sys.stderr)
    code = load_code(args.code)
    handle = DecoderHandle(code, "MSA", args.precision, args.backend)
    sim = DeviceSimulator(handle, "biawgn", args.max_iter, 0, 0x5EED1200, comm, hist_bins=args.max_iter + 1)
    s = 8 if args.precision == "f64" else 4
    bytes_per_frame_iter = s * (4 * code.E + code.n)
    default_workload = args.batch == 65536 and args.code == "1200_3_6_rand_ldpc_1" and abs(args.snr - 1.0) < 1e-9 and args.max_iter == 50

    res = run_point(sim, handle, comm, args.snr, args.steps, args.warmup, args.batch, 0, torch, not args.no_profile)
    backend_used, _ = handle.last_stats()
    extra = []
    for i, snr in enumerate(args.points):
        r = run_point(sim, handle, comm, snr, max(4, args.steps), 1, args.batch, 1 + i, torch, not args.no_profile)
        extra.append((snr, r))

    # HBM-bound reading of the same workload: the streaming backend (messages in HBM, [tile, edge, 64] layout), N = 1
    stream_res = None
    if backend_used == "fused" and comm.world == 1 and not args.no_profile:
        h2 = DecoderHandle(code, "MSA", args.precision, "stream")
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
            _lib.check(_lib.load().ldpc_count_errors(xh.data_ptr(), None, 0, it.data_ptr(), args.batch, code.n, 0, cnt.data_ptr(), st))
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
        f32_res = run_point(sim3, h3, comm, args.snr, args.steps, 1, args.batch, 0, torch, False)
        f32_res["backend"] = h3.last_stats()[0]
        del sim3, h3

    def kernel_ms(r):
        return None if not r.get("profile") else sum(r["profile"][k][0] for k in KERNEL_CLASSES)

    def summarise(snr, r, steps):
        c = r["counters"]
        frames, iter_sum = int(c[0]), int(c[3])
        fps = frames / r["seconds"]
        out = {"snr_db": snr, "frames_per_s": round(fps, 1), "ms_per_step": round(1e3 * r["seconds"] / steps, 3),
               "mean_sweeps": round(iter_sum / max(frames, 1), 3), "wer": round(int(c[1]) / max(frames, 1), 6),
               "ber": float(c[2]) / max(frames * code.n, 1),
               "algorithmic_GBps": round(iter_sum * bytes_per_frame_iter / r["seconds"] / 1e9, 1)}
        km = kernel_ms(r)
        if km is not None:
            # step time of the uninstrumented pipelined region minus the HIP-event time of the dominant kernels of the same steps
            out["kernel_ms_per_step"] = round(km / steps, 4)
            out["host_overhead_ms_per_step"] = round(1e3 * r["seconds"] / steps - km / steps, 4)
        return out

    if comm.is_root:
        head = summarise(args.snr, res, args.steps)
        c = res["counters"]
        iter_sum_rank0_share = int(c[3]) / comm.world  # the profile is rank 0's; counters are whole-job
        roof = None
        prof = res["profile"]
        if prof:
            kind = max(KERNEL_CLASSES, key=lambda k: prof[k][0])  # dominant kernel = the class with the most event time on rank 0
            ms, launches = prof[kind]
            frac_bytes = {"stream_check_pass": 2 * code.E * s, "stream_variable_pass": (2 * code.E + code.n) * s,
                          "fused_decode": bytes_per_frame_iter}[kind]
            # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes, calibrated and
            # corrected as MI355X_MICROARCH.md prescribes): collected by tools/collect_profiles.sh, committed under profiles/
            traffic = None
            if default_workload:
                key = {"stream_check_pass": "stream:%s:k_cn" % args.precision, "stream_variable_pass": "stream:%s:k_vn" % args.precision,
                       "fused_decode": "sim:%s:k_fused" % args.precision}[kind]
                for k, v in committed("hbm_traffic.json").items():
                    if k.startswith(key):
                        traffic = int(v)
            if launches > 0 and ms > 0:
                bytes_total = iter_sum_rank0_share * frac_bytes
                hbm_gbs = bytes_total / (ms * 1e-3) / 1e9
                common = {"kernel": kind, "avg_launch_ms": round(ms / launches, 4), "launches": int(launches), "traffic": traffic,
                          "all_kernels_ms": {k: round(v[0], 3) for k, v in prof.items()}}
                hbm_model = {"achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "hbm_model_frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
                             "algorithmic_bytes_per_launch": int(bytes_total / launches),
                             "note": "SURVEY 8(d) model: sum over frames of sweeps executed x %d B (%s share of s(4E+n)) / HIP-event time" % (frac_bytes, kind)}
                if kind != "fused_decode":
                    roof = dict(bound="hbm", achieved=hbm_model["achieved"], peak=HBM_PEAK_GBS, unit="GB/s", frac=hbm_model["hbm_model_frac"],
                                algorithmic_bytes_per_launch=hbm_model["algorithmic_bytes_per_launch"], note=hbm_model["note"], **common)
                else:
                    # The on-chip kernel is bound by the LDS pipe.  LDS-array cycles per frame-sweep are MEASURED (SQ_LDS_IDX_ACTIVE of
                    # this kernel on this workload / frame-sweeps of the profiled launches; profiles/lds_cycles.json); here they are
                    # multiplied by the frame-sweeps/s of the live HIP-event timing.
                    lc = next((v for k, v in committed("lds_cycles.json").items() if k.startswith("%s:k_fused" % args.precision)), None)
                    cus = torch.cuda.get_device_properties(0).multi_processor_count
                    fsps = iter_sum_rank0_share / (ms * 1e-3)
                    if lc is not None and default_workload:
                        cyc = fsps * lc["lds_idx_active_per_frame_sweep"]  # LDS-array busy cycles per second, all CUs
                        # expressed as bytes: the LDS array is 64 banks x 4 B = 256 B wide per clock and CU (MI355X_MICROARCH.md, LDS);
                        # peak at the chip's maximum clock (2.4 GHz) -- the clock the profiled launches really ran at is stated beside it
                        ach, peak = cyc * 256 / 1e9, NOMINAL_CLOCK_HZ * cus * 256 / 1e9
                        # two model figures beside the measured one (per frame-sweep, MI355X_MICROARCH.md LDS table): the conflict-free
                        # instruction minimum of the LDS array (2 cycles per gather; a stored row of 64 elements 2 cycles (4 B) / 4 (8 B)),
                        # and the same with what a store really occupies -- its address/data transfer (ds_write_addtid_b32 2 cycles per
                        # row, ds_write2st64_b64 13 per two rows) -- plus the measured conflict cycles
                        fi = handle.fused_info()
                        rows = int(fi["check_rounds"]) * int(code.row_degrees().max()) + int(fi["variable_rounds"])
                        gath = fi["lds_gather_cycles_min"]
                        arr_min = gath + rows * (4 if s == 8 else 2)
                        path = gath + rows * (6.5 if s == 8 else 2) + (lc.get("bank_conflict_per_frame_sweep") or 0)
                        avail = NOMINAL_CLOCK_HZ * cus
                        model = dict(rows_stored_per_frame_sweep=rows, gather_cycles=gath, array_cycles_conflict_free=arr_min,
                                     frac_conflict_free_minimum=round(fsps * arr_min / avail, 4),
                                     cycles_incl_store_transfer_and_conflicts=round(path, 1),
                                     frac_incl_store_transfer=round(fsps * path / avail, 4), planner_conflict_cycles=fi["conflict_cycles_planned"])
                        roof = dict(bound="lds", achieved=round(ach, 1), peak=round(peak, 1), unit="GB/s", frac=round(ach / peak, 4),
                                    instruction_model=model,
                                    lds_cycles_per_frame_sweep=lc["lds_idx_active_per_frame_sweep"],
                                    bank_conflict_cycles_per_frame_sweep=lc.get("bank_conflict_per_frame_sweep"),
                                    frame_sweeps_per_s=round(fsps, 1), peak_clock_hz=NOMINAL_CLOCK_HZ,
                                    effective_clock_hz_in_pmc_pass=lc.get("effective_clock_hz"),
                                    frac_in_pmc_pass_at_its_effective_clock=lc.get("lds_pipe_busy_in_pmc_pass"),
                                    counters_from="profiles/lds_cycles.json: " + lc.get("kernel", ""),
                                    note="LDS-array roofline: frac = busy LDS-array cycles / (CUs x 2.4 GHz), GB/s = cycles x 256 B (array width). "
                                         "Cycles per frame-sweep = SQ_LDS_IDX_ACTIVE of this kernel on this workload / frame-sweeps of the profiled "
                                         "launches (committed rocprofv3 --pmc pass), x frame-sweeps/s from the HIP events of this run; the messages "
                                         "never leave the CU, so HBM does not bound this kernel (hbm_model is informational)",
                                    hbm_model=dict(hbm_model, flag="exceeds the HBM peak: on-chip kernel, not a bound"), **common)
                    else:
                        fi = handle.fused_info()
                        roof = dict(bound="lds", achieved=None, peak=round(NOMINAL_CLOCK_HZ * cus * 256 / 1e9, 1), unit="GB/s", frac=None,
                                    frame_sweeps_per_s=round(fsps, 1), planner=fi,
                                    note="no committed SQ_LDS_IDX_ACTIVE counters for this (code, batch, SNR): run tools/collect_profiles.sh",
                                    hbm_model=dict(hbm_model, flag="on-chip kernel, not a bound"), **common)
        out = {
            "metric": ("decoded frames/s, n=1200 (3,6) min-sum max_iter=50 (+ roofline of the dominant kernel)"
                       if args.code == "1200_3_6_rand_ldpc_1" and args.max_iter == 50 else
                       "decoded frames/s, %s min-sum max_iter=%d (+ roofline of the dominant kernel)" % (args.code, args.max_iter)),
            "value": head["frames_per_s"], "unit": "frames/s", "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%s MSA over BI-AWGN, max_iter=%d, batch=%d frames/GPU, %.1f dB (mean %.2f sweeps/frame), "
                                   "all-zero word + Philox noise on device" % (args.code, args.max_iter, args.batch, args.snr, head["mean_sweeps"]),
                       "code": args.code, "n": code.n, "m": code.m, "E": code.E, "decoder": "MSA", "channel": "biawgn", "snr_db": args.snr,
                       "max_iter": args.max_iter, "batch_per_gpu": args.batch, "backend": backend_used,
                       "parallelism": "frames sharded over %d GPU(s), 1 all-reduce of counters per step, %d steps in flight" % (comm.world, sim.DEPTH)},
            "mean_sweeps": head["mean_sweeps"], "wer": head["wer"], "ber": head["ber"],
            "kernel_ms_per_step": head.get("kernel_ms_per_step"), "host_overhead_ms_per_step": head.get("host_overhead_ms_per_step"),
            "algorithmic_GBps": head["algorithmic_GBps"], "bytes_per_frame_sweep": bytes_per_frame_iter,
            "roofline": roof,
            "decode_from_hbm": hbm_leg,
            "fp32_mode": None if f32_res is None else dict(summarise(args.snr, f32_res, args.steps), backend=f32_res["backend"],
                                                           note="same workload with fp32 message arithmetic (bench.py --precision f32)"),
            "points": [summarise(snr, r, max(4, args.steps)) for snr, r in extra],
        }
        if stream_res is not None:
            sp, sc = stream_res["profile"], stream_res["counters"]
            it_sum = int(sc[3])
            legs = {}
            for kname, share in (("stream_check_pass", 2 * code.E * s), ("stream_variable_pass", (2 * code.E + code.n) * s)):
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
            out["cpu_baseline"] = cpu_baseline(code, args.snr, args.max_iter, args.precision)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    dist.finalize()


if __name__ == "__main__":
    main()
