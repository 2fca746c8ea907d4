#!/usr/bin/env python3
"""Benchmark of the BP hot path on MI355X: decoded frames/s (+ roofline of the dominant kernel, + the CPU port timed in the same run).

Default = the headline of BASELINE.json (configs[1]): code 1200_3_6_rand_ldpc_1 (the reference's own H, shipped in ldpc_decoders_amd/data/codes),
min-sum over BI-AWGN, max_iter 50, 65 536 frames per GPU, fp64 messages -- the reference's own arithmetic: hard decisions and iteration
counts bit-identical to it -- at 1.0 dB, where every frame fails and therefore executes exactly 50 sweeps (no early-exit benefit).
`--decoder MSA|SPA|BEC --channel biawgn|bsc|bec --param P --code C --precision f64|f32|f16 --batch B` select every other BASELINE
configuration through the reference's own selectors (src/main.py:11-12, src/models.py:3; `bec` pairs with the ternary erasure decoder
whatever --decoder says); `--total-batch T` fixes the frames per step over ALL GPUs (strong scaling: configs 4 / 5 state their batch for the
whole node).  tools/round_measure.sh holds the exact command of every committed line (profiles/rNN_bench*.json).

A "step" is one pass of the hot path over one batch of synthetic input, entirely on the GPU:
    channel + LLR (Philox noise keyed by the global frame index, all-zero word) -> flooding BP with the syndrome early exit of the
    reference -> bit / word error counters.

Timing.  W warm-up steps, then the timed block -- EXACTLY K steps enqueued through the pipelined driver (montecarlo.DeviceSimulator: rounds in
flight, counters all-reduced on the stream, no host sync between kernels; where the decoder takes several steps per launch -- the erasure
decoder: ldpc_simulate_rounds, one counter row per step -- up to 32 steps travel together), bracketed by barrier + torch.cuda.synchronize()
on both sides, MAX over ranks -- is repeated `--repeats` times (default 5): `ms_per_step` / `value` are the MEDIAN block, the spread is
reported beside them.  No per-kernel instrumentation runs inside a timed block; the per-kernel durations behind `roofline` come from a
SEPARATE pass of the same launches with the library's HIP events on the decode stream (`ldpc_decoder_profile`).

Roofline.  `roofline.bound` / `binding_unit` name the resource that binds the dominant kernel, `frac` = achieved / peak OF THAT UNIT:
  "lds" / "valu"  the LDS-resident kernels: busy cycles of the binding unit (LDS array, LDS issue / transfer path, VALU) / cycles available
                  at 2.4 GHz.  Cycles per frame-sweep are MEASURED counters of the very kernel that is timed (rocprofv3 --pmc on
                  tools/sim_driver.py, committed as profiles/roofline_counters.json keyed by the kernel's name) x the frame-sweeps/s of this run.
  "hbm"           the streaming kernels: bytes / HIP-event time / 8 TB/s, with the SURVEY.md 8(d) algorithmic bytes s(4E+n) for the sweep and
                  each pass's own compulsory bytes (check pass s(2E+n), variable pass s(E+2n)); PMC traffic beside it.
The 8(d) HBM-model figure of an LDS-resident kernel is kept as `hbm_model` (flagged: the messages never leave the CU, it bounds nothing).

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 one rank per GPU (RCCL) under torch.distributed.run -- started by the
caller, or, when bench.py is started as a plain process (no WORLD_SIZE), by bench.py itself: the parent makes no GPU call, starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same argv>` as a child process and relays rank 0's line (self_launch);
a node with fewer than N GPUs, or a WORLD_SIZE that is not N, ends with exit code 3 and NO line.  Frames sharded
by global frame index, ONE all-reduce of the counters per step (per block of steps); rank 0 prints ONE JSON line.  The default command's
line also carries `baseline_configs`: BASELINE configs 3-5 (sum-product / BSC, erasure decoder / BEC, n = 10 000, n = 64 800), a few steps
each through the same run_bench, each with its own `roofline` and (N = 1) `cpu_baseline`.  Every line -- N > 1
included -- carries `roofline` (rank 0's kernels), `cpu_baseline` (rank 0's host, timed BEFORE the process touches a GPU or joins the
process group: the scipy leg forks one worker per hardware thread) and `collective` (backend + the number of ranks an all-reduce of ones saw).
`run_bench(args, comm, make_handle, device)` is the whole driver layer with the decoder handle injected -- tests/test_dist_cpu.py runs it on
8 gloo ranks with a CPU stand-in for the handle; `main()` run as a script always builds the HIP handle.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
NOMINAL_CLOCK_HZ = 2.4e9  # max shader clock; the effective clock of a profiled run is in profiles/roofline_counters.json
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
    if name in ("4_2_test", "6_2_3_ldpc", "7_4_hamming", "12_3_4_ldpc"):
        return codes.get_code(name)
    return codes.load_parity_mtx(os.path.join(codes.PACKAGE_CODES_DIR, name + ".txt"))


def physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo -- os.cpu_count() counts hardware THREADS."""
    try:
        seen, phys, core = set(), None, None
        with open("/proc/cpuinfo") as fp:
            for ln in fp:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                elif not ln.strip() and phys is not None and core is not None:
                    seen.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        return len(seen) or None
    except OSError:
        return None


def resolve_workload(decoder, channel, param, snr):
    """The selectors of the reference's CLI (src/main.py:11-12, src/models.py:3) -> (algorithm of the C ABI, channel, channel parameter).
    The `bec` selector pairs with the ternary erasure decoder whatever SPA / MSA says (src/bec.py:70-125), as in the registry."""
    alg = "BEC" if (channel == "bec" or decoder == "BEC") else decoder
    channel = "bec" if alg == "BEC" else channel
    if param is None:
        param = snr if channel == "biawgn" else {"bsc": 0.07, "bec": 0.40}[channel]
    return alg, channel, float(param)


def bytes_per_frame_sweep(code, alg, precision):
    """Algorithmic bytes of one frame-sweep with the state resident in HBM.  LLR decoders: SURVEY.md 8(d), s(4E + n); fp16 storage
    (2-byte messages, 4-byte priors): 8E + 4n; erasure decoder: 2-bit messages in bit planes of 32 frames, one summary element per
    check -- (4E + m + 3n) / 4 bytes (erasure decoder)."""
    if alg == "BEC":
        return (4 * code.E + code.m + 3 * code.n) / 4.0
    if alg == "ADMM":  # one ADMM iteration with z, lambda, x, gamma, d1, d2 in HBM (the streaming kernels): x pass z, lambda, gamma in + x out;
        return 8 * (9 * code.E + 2 * code.n)  # z pass x gather, lambda in / out, z in / out, d1, d2 out; stopping test d1, d2 in (fp64)
    if alg == "ML":    # per FRAME: nothing but the counters leaves the kernel; the model prices the observation vector the search reads
        return (8 if precision == "f64" else 4) * code.n
    if precision == "f16":
        return 8 * code.E + 4 * code.n
    return (8 if precision == "f64" else 4) * (4 * code.E + code.n)


def param_label(channel, param):
    return "%.1f dB" % param if channel == "biawgn" else ("p = %g" % param if channel == "bsc" else "eps = %g" % param)


def cpu_baseline(code, alg, channel, param, max_iter, precision="f64", budget_s=10.0, scipy_leg=True):
    """CPU baselines on THIS host, on a bounded sample of the same workload (the only leg of this file that touches oracle/):
      "port"   oracle/bp_oracle.c -- a plain-C port of the reference algorithm (min-sum, sum-product, erasure decoder; sparse, so it
               also runs the codes the reference's dense H cannot hold), OpenMP over frames, every host thread;
      "scipy"  oracle/scipy_baseline.py -- per-frame scipy.sparse decoding, the reference's class of implementation, one
               process per physical host core (LLR decoders); with the calibration measured where the true reference can run
               (tests/golden/reference_timing.json: reference frames/s / scipy-baseline frames/s on identical frames) it estimates the
               reference's own rate on this host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import multiprocessing as mp

    import bp_oracle as O
    import c_oracle as C

    if alg in ("ADMM", "ML"):
        return cpu_baseline_aux(code, alg, channel, param, max_iter, budget_s)
    g = O.Edges(code.m, code.n, code.edge_chk, code.edge_var)
    cores = os.cpu_count() or 1
    dt_np = np.float64 if precision == "f64" else np.float32  # (fp16 storage has no CPU counterpart: its arithmetic is fp32)
    rng = np.random.RandomState(2024)
    zero = np.zeros(g.n, dtype=np.int64)

    def decode(nf):
        """Draw nf frames of the all-zero word through the channel (not timed), decode them (timed) -> (seconds, iters)."""
        X = np.broadcast_to(zero, (nf, g.n))
        if channel == "bec":
            y = O.bec_send(X, param, rng).astype(np.uint8)
            t0 = time.time()
            _, it = C.bec_decode(g, y, max_iter, nthreads=cores)
        elif channel == "bsc":
            y = O.bsc_send(X, param, rng)
            pri = O.bsc_priors(y, param).astype(dt_np)
            y = y.astype(dt_np)
            t0 = time.time()
            _, it = C.bp_decode(g, alg, y, pri, max_iter, dtype=dt_np, nthreads=cores)
        else:
            pri = O.biawgn_priors(O.biawgn_send(X, param, rng), param).astype(dt_np)
            t0 = time.time()
            _, it = C.bp_decode(g, alg, None, pri, max_iter, dtype=dt_np, nthreads=cores)
        return time.time() - t0, it

    mem_frames = max(cores, int(2e9 // (16 * g.n)))                  # the sample's arrays stay below ~2 GB whatever n is
    probe = max(cores, min(64 * cores, 20_000_000 // g.n, mem_frames))
    rate = probe / max(decode(probe)[0], 1e-6)
    nf = int(max(probe, min(rate * budget_s, 400000, mem_frames)))
    dt, it = decode(nf)
    arith = "erasure decoder (integer)" if alg == "BEC" else ("fp64" if precision == "f64" else "fp32")
    out = {"value": round(nf / dt, 1), "unit": "frames/s", "cores": cores, "physical_cores": physical_cores(), "kind": "port",
           "sample": "%d frames, same H / %s over %s at %s / max_iter %d, %s C port of the reference algorithm (oracle/bp_oracle.c), "
                     "%d OpenMP threads, %.1f s, mean %.1f sweeps/frame" % (nf, alg, channel, param_label(channel, param), max_iter, arith, cores, dt,
                                                                             float(it.mean()))}
    if alg == "BEC":
        out["scipy"] = {"skipped": "the scipy.sparse leg restates the LLR decoders (src/bpa.py); the erasure decoder's CPU figure is the C port"}
        return out
    if not scipy_leg:
        out["scipy"] = {"skipped": "baseline_configs entry: C port only (the scipy leg runs on the configuration's own bench line)"}
        return out
    # per-frame scipy.sparse baseline, one process per core, each decoding its own frame stream for about budget_s seconds
    try:
        # one single-thread process per PHYSICAL core ("one process per host core"; with one per hardware thread two processes
        # share a core and the per-process rate is not the reference's); forked BEFORE this process initialises a GPU runtime
        procs = physical_cores() or cores
        task = (code.m, code.n, code.edge_chk, code.edge_var, alg, channel, param, max_iter)
        with mp.get_context("fork").Pool(procs) as pool:
            # a short pass of every process sizes the sample (the rate per process UNDER LOAD, not that of one process alone)
            pool.map(_scipy_probe, [task + (1, 10 + i) for i in range(procs)])  # imports, warm-up
            t0 = time.time()
            pool.map(_scipy_probe, [task + (2, 50 + i) for i in range(procs)])
            per = 2.0 / max(time.time() - t0, 1e-3)
            frames_each = max(2, int(per * budget_s))
            t0 = time.time()
            res = pool.map(_scipy_probe, [task + (frames_each, 100 + i) for i in range(procs)])
            wall = time.time() - t0
        frames = sum(r["frames"] for r in res)
        sc = {"value": round(frames / wall, 2), "unit": "frames/s", "cores": procs, "physical_cores": physical_cores(), "kind": "scipy",
              "per_core_frames_per_s": round(frames / wall / procs, 3),
              "sample": "%d frames (%d per process), same H / %s over %s at %s / max_iter %d, fp64 per-frame scipy.sparse decoder "
                        "(oracle/scipy_baseline.py), %d single-thread processes (one per physical core), %.1f s, mean %.1f sweeps/frame" % (
                            frames, frames_each, alg, channel, param_label(channel, param), max_iter, procs, wall,
                            sum(r["iters"] for r in res) / max(frames, 1))}
        with open(os.path.join(ROOT, "tests", "golden", "reference_timing.json")) as fp:
            tj = json.load(fp)
        for pt in tj["points"]:
            if channel == "biawgn" and code.n == 1200 and pt["decoder"] == alg and abs(pt["snr_db"] - param) < 1e-9:
                sc.update(calibration_reference_over_scipy=round(pt["calibration"], 4),
                          calibration_measured_on="%s, %d frames: reference %.2f frames/s, scipy baseline %.2f frames/s" % (
                              tj["host"], pt["frames"], pt["frames_per_s"], pt["baseline_frames_per_s"]),
                          reference_estimate_frames_per_s=round(frames / wall * pt["calibration"], 2),
                          reference_estimate_note="scipy-baseline rate on this host x calibration: what the reference's own Python would "
                                                  "reach here with one process per physical core (it is single-threaded)")
        out["scipy"] = sc
    except Exception as e:  # the baseline is a report, never a reason to lose the benchmark line
        out["scipy"] = {"error": repr(e)}
    return out


def cpu_baseline_aux(code, alg, channel, param, max_iter, budget_s):
    """CPU baselines of the SURVEY 8(f) decoders: ADMM = oracle/admm_oracle.c (OpenMP over frames, every host thread; its projection is pinned
    to the reference's own projection.cpp), ML = oracle/ml_oracle.py (numpy, the reference's expressions, ONE core)."""
    import bp_oracle as O

    rng = np.random.RandomState(2024)
    cores = os.cpu_count() or 1
    zero = np.zeros(code.n, dtype=np.int64)
    if alg == "ADMM":
        import admm_oracle as A

        class G:
            m, n, chk, var = code.m, code.n, code.edge_chk, code.edge_var

        def decode(nf):
            gamma = O.biawgn_priors(O.biawgn_send(np.broadcast_to(zero, (nf, code.n)), param, rng), param)
            t0 = time.time()
            _, it, _ = A.admm_decode(G, gamma, 3.0, 1e-5, max_iter)
            return time.time() - t0, it

        probe = max(cores, min(4 * cores, 2_000_000 // code.n))
        rate = probe / max(decode(probe)[0], 1e-6)
        nf = int(max(probe, min(rate * budget_s, 200000)))
        dt, it = decode(nf)
        return {"value": round(nf / dt, 1), "unit": "frames/s", "cores": cores, "physical_cores": physical_cores(), "kind": "port",
                "sample": "%d frames, same H / ADMM (mu 3, eps 1e-5) over %s at %s / max_iter %d, fp64 C port of the reference algorithm incl. its "
                          "parity-polytope projection (oracle/admm_oracle.c), %d OpenMP threads, %.1f s, mean %.1f iterations/frame" % (
                              nf, channel, param_label(channel, param), max_iter, cores, dt, float(it.mean())),
                "scipy": {"skipped": "the scipy.sparse leg restates the BP decoders (src/bpa.py)"}}
    import ml_oracle as M
    from ldpc_decoders_amd.models import models

    chan = models[channel].Channel(param)
    coef = M.ml_coefficients(channel, param)
    np.random.seed(1)
    nf, t0 = 0, time.time()
    with np.errstate(all="ignore"):
        while time.time() - t0 < min(budget_s, 5.0):
            M.ml_decode(channel, code.cb, chan.send(zero), coef)
            nf += 1
    dt = time.time() - t0
    return {"value": round(nf / dt, 1), "unit": "frames/s", "cores": 1, "physical_cores": physical_cores(), "kind": "port",
            "sample": "%d frames, exhaustive search over the %d codewords of %s over %s at %s, numpy restatement of the reference's expressions "
                      "(oracle/ml_oracle.py), one core, %.1f s (channel draw included)" % (nf, len(code.cb), "the code", channel, param_label(channel, param), dt),
            "scipy": {"skipped": "the scipy.sparse leg restates the BP decoders (src/bpa.py)"}}


class AuxHandle:
    """What run_bench asks of a decoder handle, for the SURVEY 8(f) decoders (ADMM: _device.AdmmHandle, ML: _device.MlHandle)."""

    def __init__(self, code, alg, precision, channel):
        from ldpc_decoders_amd import _device

        self.alg, self.code = alg, code
        if alg == "ADMM":
            self.inner = _device.AdmmHandle(code)
            self.inner.mu, self.inner.eps = 3.0, 1e-5
        else:
            self.inner = _device.MlHandle(code.cb, channel, precision)

    def simulate(self, *a, **k):
        return self.inner.simulate(*a, **k)

    def last_stats(self):
        return ("admm-" + self.inner.last_backend() if self.alg == "ADMM" else "ml"), 0

    def kernel_name(self, simulate=False):
        if self.alg == "ADMM":
            rows = (self.code.m + 63) // 64  # the shape ldpc_admm.hip picks: 4 or 8 waves, one or two passes of the check phase
            lds = "k_admm_lds<6, 3, 2, 4, 1>" if rows <= 4 else ("k_admm_lds<6, 3, 2, 8, 1>" if rows <= 8 else "k_admm_lds<6, 3, 3, 8, 2>")
            return lds if self.inner.last_backend() == "lds" else "k_admm_z_fixed<6>"
        obs = {"f64": "double", "f32": "float"}[self.inner.precision] if self.inner.channel == "biawgn" else "unsigned char"
        return "k_ml<%s, %d>" % (obs, {"biawgn": 0, "bsc": 1, "bec": 2}[self.inner.channel])

    def timed_launches(self, channel, param, frames, max_iter, steps, torch):
        """HIP-event time of `steps` launches of the dominant kernel ALONE on torch's current stream (the library launches on it), with
        inputs resident in HBM -> (ms per launch, frame-iterations [ADMM] or frames [ML] per launch)."""
        from ldpc_decoders_amd import _lib

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if self.alg == "ADMM":
            gamma = torch.empty((frames, self.code.n), dtype=torch.float64, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.load().ldpc_channel(_lib.CHANNEL[channel], _lib.DTYPE["f64"], float(param), 0, 0x5EED1200, 9, 0, frames, self.code.n,
                                                gamma.data_ptr(), None, st))
            _, it, _ = self.inner.decode_device(gamma, 3.0, 1e-5, max_iter)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                self.inner.decode_device(gamma, 3.0, 1e-5, max_iter)
            e1.record()
            torch.cuda.synchronize()
            # the estimate buffers of decode_device are torch allocations from the caching allocator: no device synchronisation in the loop
            return e0.elapsed_time(e1) / steps, float(it.sum().item()) + frames  # (the iteration that meets the test counts: it + 1 per frame)
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        self.inner.simulate(channel, param, 0, 0x5EED1200, 9, 0, frames, 0, cnt)
        torch.cuda.synchronize()
        e0.record()
        for i in range(steps):
            self.inner.simulate(channel, param, 0, 0x5EED1200, 9, i * frames, frames, 0, cnt)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps, float(frames)


def _scipy_probe(task):
    m, n, chk, var_idx, alg, channel, param, max_iter, frames, seed = task
    os.environ["OMP_NUM_THREADS"] = "1"
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from scipy_baseline import ScipyBP

    dec = ScipyBP(m, n, chk, var_idx, alg, max_iter)
    rng = np.random.RandomState(seed)
    var = 10 ** (-param / 10)
    llr = np.log(1 - param) - np.log(param) if channel == "bsc" else 0.0
    t0 = time.time()
    iters = 0
    for _ in range(frames):
        if channel == "bsc":  # src/bsc.py:16,21-25
            y = (rng.random(n) < param).astype(np.int64)
            dec.decode(y, llr * (1 - 2 * y))
        else:                 # src/biawgn.py:17-18,28
            y = -1 + rng.normal(0, np.sqrt(var), n)
            dec.decode(y, -2 * y / var)
        iters += dec.iterations
    dt = time.time() - t0
    return {"frames": frames, "iters": iters, "frames_per_s": frames / max(dt, 1e-9)}


def _sync(device, torch):
    if device == "cuda":
        torch.cuda.synchronize()


def run_point(sim, handle, comm, snr, steps, warmup, per_round, stream_id, torch, kernel_pass=True, repeats=1, device="cuda"):
    """`repeats` timed blocks of `steps` steps at one channel parameter (uninstrumented, pipelined), then -- separately -- the steps of one
    block again with the library's HIP-event kernel timing switched on.  `per_round` = frames of one step over ALL ranks (each rank
    decodes its shard of the global frame range).  Returns dict(seconds (median block), blocks, counters (one block), profile)."""
    frame0 = 0
    if kernel_pass:
        handle.set_profiling(False)
    for _ in range(warmup):
        sim.run_round(snr, stream_id, frame0, per_round)
        frame0 += per_round
    first = frame0
    blocks, tot = [], None
    rpl = sim.rounds_per_launch() if hasattr(sim, "rounds_per_launch") else 1
    for _rep in range(max(1, repeats)):
        frame0 = first  # every block decodes the same frames: identical work, identical counters
        tot = np.zeros(4 + sim.hist_bins, dtype=np.int64)
        comm.barrier()
        _sync(device, torch)
        t0 = time.perf_counter()
        inflight = []
        if rpl > 1:
            # the erasure decoder's rounds are 0.2 ms launches: up to `rpl` steps travel in ONE launch, each step with its own counter row
            # (ldpc_simulate_rounds) -- the same K steps, the same frames, the same per-step counters
            left = steps
            while left > 0:
                if len(inflight) == sim.DEPTH:
                    tot += sim.finish_rounds(inflight.pop(0)).sum(axis=0)
                r = min(rpl, left)
                inflight.append(sim.launch_rounds(snr, stream_id, frame0, per_round, r))
                frame0 += per_round * r
                left -= r
            while inflight:
                tot += sim.finish_rounds(inflight.pop(0)).sum(axis=0)
        for _ in range(steps if rpl == 1 else 0):
            if len(inflight) == sim.DEPTH:
                tot += sim.finish_round(inflight.pop(0))
            inflight.append(sim.launch_round(snr, stream_id, frame0, per_round))
            frame0 += per_round
        while inflight:
            tot += sim.finish_round(inflight.pop(0))
        _sync(device, torch)
        comm.barrier()
        blocks.append(comm.max_float(time.perf_counter() - t0))
    prof = None
    if kernel_pass:  # same frames again, one step at a time, with HIP events around the dominant kernels (rank-local)
        handle.set_profiling(True)
        handle.read_profile(reset=True)
        f = first
        left = steps
        while left > 0:  # the same launches as the timed region (several steps per launch where the decoder takes them), one at a time
            r = min(rpl, left)
            if rpl > 1:
                sim.finish_rounds(sim.launch_rounds(snr, stream_id, f, per_round, r))
            else:
                sim.run_round(snr, stream_id, f, per_round)
            f += per_round * r
            left -= r
        prof = handle.read_profile(reset=True)
        handle.set_profiling(False)
    return dict(seconds=statistics.median(blocks), blocks=blocks, counters=tot, profile=prof)


def committed(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fp:
            return json.load(fp)
    except Exception:
        return {}


_CODE_HASHES = None


def loaded_kernel_hash(kernel_name):
    """Hash of the machine code of `kernel_name` in the library this process decodes with (tools/kernel_resources.py: file parsing only)."""
    global _CODE_HASHES
    if _CODE_HASHES is None:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import kernel_resources
            from ldpc_decoders_amd import _lib

            _CODE_HASHES = kernel_resources.kernel_code_hashes(_lib.library_path())
        except Exception as e:  # the check is evidence hygiene, never a reason to lose the line
            _CODE_HASHES = {"error": repr(e)}
    return _CODE_HASHES.get(kernel_name)


def counters_freshness(entry, kernel_name):
    """-> (stale, note).  stale True: the counters were collected on ANOTHER body of this kernel (same name, different machine code) -- the
    line must not price the new timing with the old cycles; None: no hash to compare (entry older than round 6, or the library unreadable)."""
    want, have = entry.get("kernel_code_sha"), loaded_kernel_hash(kernel_name)
    if not want or not have:
        return None, "unverified: %s" % ("the counters entry carries no kernel_code_sha" if not want else "no code hash for the loaded library (%s)" % _CODE_HASHES.get("error"))
    if want != have:
        return True, "STALE: counters collected on kernel code %s (lib %s, HEAD %s), the loaded library holds %s -- re-run tools/collect_rooflines.sh" % (
            want, (entry.get("lib_sha256") or "?")[:12], (entry.get("head") or "?")[:12], have)
    return False, "kernel code %s == the code the counters were collected on (HEAD %s)" % (have, (entry.get("head") or "?")[:12])


def fused_roofline(kernel_name, frame_sweeps_per_s, cus, counters=None):
    """LDS / VALU roofline of an LDS-resident kernel from its committed PMC counters (profiles/roofline_counters.json, keyed by kernel
    name): busy cycles per frame-sweep x frame-sweeps/s / (CUs x [4 SIMDs x] 2.4 GHz).  None when the kernel has no committed counters.
    Every entry names the machine code it was measured on (`kernel_code_sha`); against another body of the same kernel `frac` is null and
    `counters_stale` true."""
    counters = committed("roofline_counters.json") if counters is None else counters
    e = counters.get(kernel_name)
    if not e:
        return None
    stale, fresh_note = counters_freshness(e, kernel_name)
    if stale:
        return dict(bound="lds", binding_unit=None, frac=None, achieved=None, peak=round(NOMINAL_CLOCK_HZ * cus * 256 / 1e9, 1), unit="GB/s",
                    counters_stale=True, counters_check=fresh_note, frame_sweeps_per_s=round(frame_sweeps_per_s, 1), counters_kernel=kernel_name)
    lds = frame_sweeps_per_s * e["lds_idx_active_per_frame_sweep"] / (cus * NOMINAL_CLOCK_HZ)
    valu = frame_sweeps_per_s * e["valu_active_cycles_per_frame_sweep"] / (cus * 4 * NOMINAL_CLOCK_HZ)
    array_peak = NOMINAL_CLOCK_HZ * cus * 256 / 1e9  # the LDS array is 64 banks x 4 B wide per clock and CU
    useful = frame_sweeps_per_s * (e["lds_idx_active_per_frame_sweep"] - e["bank_conflict_per_frame_sweep"]) / (cus * NOMINAL_CLOCK_HZ)
    # the unit of the LDS pipeline that binds an 8-byte-element kernel is the store (issue / transfer) path, not the array the counter above
    # sees: measured load / store instruction counts x per-instruction cycles -- the hardware guide's (2 / 6) and the ones measured on this
    # chip (2.55 / 6.3, additive in a mixed stream: profiles/r04_lds_store_path.txt)
    path = {}
    for key, name in (("lds_path_cycles_per_frame_sweep", "lds_store_path_frac_guide_constants"),
                      ("lds_path_cycles_per_frame_sweep_measured_constants", "lds_store_path_frac_measured_constants")):
        if e.get(key):
            path[name] = round(frame_sweeps_per_s * e[key] / (cus * NOMINAL_CLOCK_HZ), 4)
    # `frac` is ONE thing: busy cycles of the BINDING unit / cycles available at 2.4 GHz.  Candidates: the LDS array (SQ_LDS_IDX_ACTIVE), the
    # LDS issue / transfer path (instruction counts x the hardware guide's per-instruction cycles; the constants measured on this chip stay
    # a second field), the VALU (issue model on the measured instruction mix).  `achieved` / `peak` are that unit's, so frac == achieved / peak.
    units = [("lds_array", lds), ("valu", valu)]
    if "lds_store_path_frac_guide_constants" in path:
        units.append(("lds_store_path", path["lds_store_path_frac_guide_constants"]))
    binding, frac = max(units, key=lambda t: t[1])
    if binding == "lds_array":
        achieved, peak, unit = lds * array_peak, array_peak, "GB/s"
    elif binding == "lds_store_path":
        peak = cus * NOMINAL_CLOCK_HZ / 1e9
        achieved, unit = frac * peak, "G cycles/s of the LDS issue/transfer path (all CUs)"
    else:
        peak = cus * 4 * NOMINAL_CLOCK_HZ / 1e9
        achieved, unit = frac * peak, "G VALU issue cycles/s (all SIMDs)"
    return dict(bound="valu" if binding == "valu" else "lds", binding_unit=binding, frac=round(frac, 4), achieved=round(achieved, 1), peak=round(peak, 1), unit=unit,
                lds_frac=round(lds, 4), lds_array_GBps=round(lds * array_peak, 1), lds_array_peak_GBps=round(array_peak, 1),
                lds_frac_without_bank_conflicts=round(useful, 4), valu_frac=round(valu, 4), **path,
                lds_cycles_per_frame_sweep=e["lds_idx_active_per_frame_sweep"], bank_conflict_cycles_per_frame_sweep=e["bank_conflict_per_frame_sweep"],
                valu_busy_cycles_per_frame_sweep=e["valu_active_cycles_per_frame_sweep"], valu_insts_per_frame_sweep=e["insts_valu_per_frame_sweep"],
                lds_insts_per_frame_sweep=e["insts_lds_per_frame_sweep"], frame_sweeps_per_s=round(frame_sweeps_per_s, 1),
                peak_clock_hz=NOMINAL_CLOCK_HZ, effective_clock_hz_in_pmc_pass=e.get("effective_clock_hz_in_pmc_pass"),
                lds_busy_frac_in_pmc_pass=e.get("lds_busy_frac_in_pmc_pass"), valu_busy_frac_in_pmc_pass=e.get("valu_busy_frac_in_pmc_pass"),
                wave_time_shares=dict(waiting=e.get("wait_any_share"), issue_stall=e.get("wait_inst_any_share"), issuing=e.get("active_inst_any_share")),
                counters_kernel=kernel_name, counters_workload=e.get("workload"), counters_from=e.get("counters_from"),
                counters_stale=stale, counters_check=fresh_note)


def run_bench(args, comm, make_handle=None, device="cuda", cpu_base=None):
    """The benchmark driver: returns the result dict on rank 0 (None elsewhere).  `make_handle(code, alg, precision, backend)` builds
    the decoder handle (default: the HIP DecoderHandle -- no CPU path exists in the product).  `cpu_base`: rank 0's CPU baseline,
    measured by the caller before any GPU / process-group initialisation (main() does)."""
    import torch

    # the collective the counters travel through, proven on the line itself: an all-reduce of ones must see every rank
    ranks_seen = int(np.asarray(comm.all_reduce_sum(np.ones(1, dtype=np.int64)))[0]) if comm.group else 1
    collective = {"backend": ("rccl (torch.distributed 'nccl')" if comm.backend == "nccl" else comm.backend) if comm.group else None,
                  "ranks_seen": ranks_seen, "op": "all_reduce(sum) of the int64 counters, once per step"}

    from ldpc_decoders_amd.montecarlo import DeviceSimulator

    code = load_code(args.code)
    alg, channel, param = resolve_workload(args.decoder, args.channel, args.param, args.snr)
    if make_handle is None and alg not in ("ADMM", "ML"):
        from ldpc_decoders_amd._device import DecoderHandle as make_handle  # noqa: N813
    aux = alg in ("ADMM", "ML")
    precision = "f32" if alg == "BEC" else ("f64" if alg == "ADMM" else args.precision)  # the erasure decoder has no floating-point state (2-bit messages in bit planes)
    if aux and make_handle is None:
        make_handle = lambda code_, alg_, prec_, backend_: AuxHandle(code_, alg_, prec_, channel)  # noqa: E731
    points = args.points if args.points is not None else {"biawgn": [2.0, 3.0], "bsc": [0.05], "bec": [0.35]}[channel]
    hist_bins = min(args.max_iter + 1, 60)  # the in-kernel histogram has 60 bins: sweeps >= 59 share the last one (reported on the line)
    msa_biawgn = alg == "MSA" and channel == "biawgn"
    if aux and args.points is None:
        points = []
    handle = make_handle(code, alg, precision, args.backend)
    sim = DeviceSimulator(handle, channel, args.max_iter, 0, 0x5EED1200, comm, hist_bins=hist_bins, device=device)
    s = 8 if precision == "f64" else 4
    bytes_per_frame_iter = bytes_per_frame_sweep(code, alg, precision)  # SURVEY.md 8(d) for the LLR decoders
    f16 = precision == "f16"
    side_legs = not args.no_profile and comm.world == 1 and device == "cuda" and not getattr(args, "headline_only", False) and not aux
    kernel_pass = not args.no_profile and device == "cuda" and not aux
    # frames of one step over all ranks: weak scaling (--batch frames per GPU, the default) or strong (--total-batch frames per step
    # whatever N is -- BASELINE configs 4 and 5 state their batch for the whole 8-GPU node; a total that N does not divide is split
    # as evenly as possible by Comm.shard)
    strong = args.total_batch is not None
    per_round = int(args.total_batch) if strong else args.batch * comm.world
    rank_batch = comm.shard(0, per_round)[1]  # this rank's frames per step (rank 0: the largest shard)

    res = run_point(sim, handle, comm, param, args.steps, args.warmup, per_round, 0, torch, kernel_pass, args.repeats, device)
    backend_used, _ = handle.last_stats()
    extra = []
    for i, snr in enumerate(points):
        r = run_point(sim, handle, comm, snr, max(4, args.steps), 1, per_round, 1 + i, torch, kernel_pass, 1, device)
        extra.append((snr, r))

    # HBM-bound reading of the same workload: the streaming backend (state resident in HBM, [tile, edge, 64] layout), N = 1
    stream_res = None
    if backend_used == "fused" and side_legs and msa_biawgn:
        h2 = make_handle(code, "MSA", precision, "stream")
        sim2 = DeviceSimulator(h2, "biawgn", args.max_iter, 0, 0x5EED1200, comm, hist_bins=hist_bins)
        stream_res = run_point(sim2, h2, comm, param, 2, 1, per_round, 0, torch)
        del sim2, h2

    # Same frames with the channel output RESIDENT IN HBM when the timed region starts (channel kernel run beforehand): decode + count
    # only, rank 0's shard.  Reported beside `value` (which times the whole hot path: channel + decode + count).
    hbm_leg = None
    if side_legs:
        from ldpc_decoders_amd import _lib

        pri, y0 = handle.channel_device(channel, param, 0, 0x5EED1200, 0, 0, rank_batch)
        xh, it = handle.decode_device(pri, y0, args.max_iter)
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            handle.decode_device(pri, y0, args.max_iter, xhat=xh, iters=it)
            _lib.check(_lib.load().ldpc_count_errors(xh.data_ptr(), None, 0, it.data_ptr(), rank_batch, code.n, 0, cnt.data_ptr(), st))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        hbm_leg = {"frames_per_s": round(args.steps * rank_batch / dt, 1), "ms_per_step": round(1e3 * dt / args.steps, 3),
                   "mean_sweeps": round(float(cnt[3]) / float(cnt[0]), 3),
                   "note": "%s resident in HBM before the timed region; decode (ldpc_decode) + error counting (ldpc_count_errors)" % (
                       "received symbols [B,n] u8" if alg == "BEC" else "priors [B,n] %s%s" % (precision, " + received word [B,n] u8" if y0 is not None else ""))}
        del pri, y0, xh, it

    # the fp32 throughput mode of the same workload (statistically identical curves, not bit-identical frame by frame), N = 1
    f32_res = None
    if precision == "f64" and side_legs and msa_biawgn:
        h3 = make_handle(code, "MSA", "f32", args.backend)
        sim3 = DeviceSimulator(h3, "biawgn", args.max_iter, 0, 0x5EED1200, comm, hist_bins=hist_bins)
        f32_res = run_point(sim3, h3, comm, param, args.steps, 1, per_round, 0, torch, True, 3)
        f32_res["backend"] = h3.last_stats()[0]
        f32_res["kernel"] = h3.kernel_name(True) if f32_res["backend"] == "fused" else ""
        del sim3
        # exact-in-fp32 mode: the same fp32 kernel family on priors rounded to multiples of 2^-8, under the exactness guard; frames beyond
        # it are decoded again in fp64 -- every counted frame is what the fp64 reference returns for those priors (never `value`)
        exact_res = None
        try:
            sim4 = DeviceSimulator(h3, "biawgn", args.max_iter, 0, 0x5EED1200, comm, hist_bins=hist_bins, prior_grid=8)
            exact_steps = max(24, args.steps)  # whole blocks of eight guarded launches (one fp64 redo pass per block)
            exact_res = run_point(sim4, h3, comm, param, exact_steps, 1, per_round, 0, torch, False, 1)
            exact_res["redone"], exact_res["steps"] = sim4.redone, exact_steps
            exact_res["depth"] = sim4.pipeline_depth()
            del sim4
        except Exception as e:  # a side leg must not cost the benchmark line
            exact_res = {"error": repr(e)}
        del h3

    # fp16 STORAGE mode of the streaming kernels (2-byte messages, fp32 arithmetic): the same workload, where the headline runs on the
    # streaming kernels in fp32 (codes that do not fit the LDS: config 5).  A tolerance mode -- its own block, never `value`.
    f16_res = None
    if backend_used == "stream" and precision == "f32" and side_legs and msa_biawgn:
        h5 = make_handle(code, "MSA", "f16", "stream")
        sim5 = DeviceSimulator(h5, "biawgn", args.max_iter, 0, 0x5EED1200, comm, hist_bins=hist_bins)
        f16_res = run_point(sim5, h5, comm, param, args.steps, 1, per_round, 0, torch, True, 1)
        del sim5, h5

    def kernel_ms(r):
        return None if not r.get("profile") else sum(r["profile"][k][0] for k in KERNEL_CLASSES)

    def summarise(snr, r, steps):
        c = r["counters"]
        frames, iter_sum = int(c[0]), int(c[3])
        fps = frames / r["seconds"]
        per_step = [1e3 * b / steps for b in r["blocks"]]
        out = {("snr_db" if channel == "biawgn" else "param"): snr, "frames_per_s": round(fps, 1), "ms_per_step": round(1e3 * r["seconds"] / steps, 4),
               "ms_per_step_min": round(min(per_step), 4), "ms_per_step_max": round(max(per_step), 4), "timed_blocks": len(per_step),
               "mean_sweeps": round(iter_sum / max(frames, 1), 3), "wer": round(int(c[1]) / max(frames, 1), 6),
               "ber": float(c[2]) / max(frames * code.n, 1),
               "algorithmic_GBps": round(iter_sum * bytes_per_frame_iter / r["seconds"] / 1e9, 1)}
        km = kernel_ms(r)
        if km is not None:
            # step time of the uninstrumented pipelined region minus the HIP-event time of the kernels of the same steps
            out["kernel_ms_per_step"] = round(km / steps, 4)
            total = r["profile"].get("stream_decode_total", (0, 0))[0]
            if total > 0:  # streaming backend: the whole decode, side kernels (tile load, syndrome, repack, unpack) included
                out["decode_ms_per_step"] = round(total / steps, 4)
                out["side_kernels_ms_per_step"] = round((total - km) / steps, 4)
                out["host_overhead_ms_per_step"] = round(1e3 * r["seconds"] / steps - total / steps, 4)
                out["host_overhead_note"] = "step - HIP-event time of the whole decode; what is left is the channel and counting kernels of the step + host gaps"
            else:
                out["host_overhead_ms_per_step"] = round(1e3 * r["seconds"] / steps - km / steps, 4)
        return out

    if not comm.is_root:
        return None
    head = summarise(param, res, args.steps)
    c = res["counters"]
    iter_sum_rank0_share = int(c[3]) / comm.world  # the profile is rank 0's; counters are whole-job
    roof = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
            "note": "no per-kernel HIP-event pass in this run (--no-profile, or not on a GPU)"}
    prof = res["profile"]
    cus = torch.cuda.get_device_properties(0).multi_processor_count if device == "cuda" else 256
    if prof:
        kind = max(KERNEL_CLASSES, key=lambda k: prof[k][0])  # dominant kernel = the class with the most event time on rank 0
        ms, launches = prof[kind]
        if launches > 0 and ms > 0:
            common = {"kernel_class": kind, "avg_launch_ms": round(ms / launches, 4), "launches": int(launches),
                      "all_kernels_ms": {k: round(v[0], 3) for k, v in prof.items()}}
            if kind == "fused_decode":
                kname = handle.kernel_name(True)
                fsps = iter_sum_rank0_share / (ms * 1e-3)
                hbm_gbs = iter_sum_rank0_share * bytes_per_frame_iter / (ms * 1e-3) / 1e9
                hbm_model = {"achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "hbm_model_frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
                             "flag": "exceeds the HBM peak: on-chip kernel, not a bound",
                             "note": "SURVEY 8(d) model: sum over frames of sweeps executed x %d B / HIP-event time" % bytes_per_frame_iter}
                roof = fused_roofline(kname, fsps, cus)
                traffic = None
                for k, v in committed("roofline_counters.json").items():
                    if k.startswith("hbm:") and k.endswith(":" + kname):
                        traffic = v["hbm_bytes_per_launch"]
                if roof is None:
                    roof = dict(bound="lds", frac=None, achieved=None, peak=round(NOMINAL_CLOCK_HZ * cus * 256 / 1e9, 1), unit="GB/s",
                                frame_sweeps_per_s=round(fsps, 1), note="no committed PMC counters for %s: run tools/collect_rooflines.sh" % kname)
                roof.update(kernel=kname, traffic=traffic, hbm_model=hbm_model,
                            note="LDS-resident kernel: frac = busy cycles of the binding unit / available cycles at 2.4 GHz.  Cycles per frame-sweep are "
                                 "PMC counters of THIS kernel (the simulate variant that is timed): SQ_LDS_IDX_ACTIVE; VALU = an issue model on the measured instruction mix "
                                 "(2 cycles per wave64 instruction, 4 per fp64 add/mul/fma, 8 per transcendental), "
                                 "committed under profiles/, x the frame-sweeps/s of the HIP-event timing of this run; `traffic` = PMC HBM bytes per launch",
                            **common)
            else:
                # streaming kernels: each pass priced with its own compulsory bytes; the sweep with the section-8(d) model
                share = {"stream_check_pass": (2 * code.E + code.n) * s, "stream_variable_pass": (code.E + 2 * code.n) * s}
                if f16:
                    share = {"stream_check_pass": 4 * code.E, "stream_variable_pass": 4 * code.E + 4 * code.n}
                if alg == "BEC":  # bit planes: check pass E reads + m summary writes, variable pass 3E + 3n, 8-byte elements per 32 frames
                    share = {"stream_check_pass": (code.E + code.m) / 4.0, "stream_variable_pass": (3 * code.E + 3 * code.n) / 4.0}
                legs = {}
                for kname in ("stream_check_pass", "stream_variable_pass"):
                    kms, kl = prof[kname]
                    if kl:
                        gbs = iter_sum_rank0_share * share[kname] / (kms * 1e-3) / 1e9
                        legs[kname] = {"achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "avg_launch_ms": round(kms / kl, 4),
                                       "launches": int(kl), "compulsory_bytes_per_frame_sweep": share[kname]}
                pair_ms = prof["stream_check_pass"][0] + prof["stream_variable_pass"][0]
                sweep_gbs = iter_sum_rank0_share * bytes_per_frame_iter / (pair_ms * 1e-3) / 1e9
                traffic = None
                pref = ("k_becs_cn" if kind == "stream_check_pass" else "k_becs_vn<") if alg == "BEC" else \
                       ("k_cn16<" if kind == "stream_check_pass" else "k_vn16<") if f16 else \
                       ("k_cn<" if kind == "stream_check_pass" else "k_vn<") + ("double" if s == 8 else "float")
                traffic_note = None
                for k, v in committed("roofline_counters.json").items():
                    if k.startswith("hbm:") and (":" + pref) in k and v.get("workload", "").startswith(args.code + " "):
                        traffic = v["hbm_bytes_per_launch"]
                        # the counters were collected on launches of another batch (a full-size PMC pass of n = 64 800 takes minutes): the
                        # streaming kernels move bytes in proportion to the frames of a launch, so the figure is scaled to this run's batch
                        try:
                            cb = int(v["workload"].split(" batch ")[1].split()[0])
                            if cb != rank_batch:
                                traffic = int(traffic * rank_batch / cb)
                                traffic_note = "PMC bytes per launch of %d frames x %d / %d" % (cb, rank_batch, cb)
                        except (IndexError, ValueError):
                            pass
                roof = dict(bound="hbm", achieved=round(sweep_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(sweep_gbs / HBM_PEAK_GBS, 4),
                            algorithmic_bytes_per_frame_sweep=bytes_per_frame_iter, passes=legs, traffic=traffic, traffic_scaled=traffic_note,
                            kernel="k_becs_cn + k_becs_vn (one sweep)" if alg == "BEC" else "k_cn16 + k_vn16 (one sweep)" if f16 else "k_cn + k_vn (one sweep)",
                            note="streaming backend: achieved = executed frame-sweeps x algorithmic bytes (SURVEY 8(d): s(4E+n); fp16 storage 8E+4n; "
                                 "erasure bit planes (4E+m+3n)/4) / HIP-event time of the two passes; "
                                 "`passes` prices each kernel with its own compulsory bytes (check pass: c2v in + out + each marginal once = s(2E+n); "
                                 "variable pass: c2v in + prior in + marginal out = s(E+2n)); `traffic` = PMC HBM bytes per launch of the dominant "
                                 "pass on this code (profiles/roofline_counters.json)", **common)
    if prof is not None and roof.get("frac") is None and device == "cuda" and backend_used == "stream":
        # a streaming decode without per-pass HIP events: the WHOLE step (channel + every kernel + host gaps) against the HBM peak
        gbs = head["algorithmic_GBps"] / comm.world
        roof = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None,
                    algorithmic_bytes_per_frame_sweep=bytes_per_frame_iter, kernel="whole step",
                    note="no per-pass HIP events on this backend: executed frame-sweeps x algorithmic bytes / WALL time of the step")
    if aux and device == "cuda" and not args.no_profile and hasattr(handle, "timed_launches"):
        # the dominant kernel alone (k_admm_lds / k_ml), HIP events on the stream it is launched on, inputs resident in HBM
        ms, units = handle.timed_launches(channel, param, rank_batch, args.max_iter, max(2, min(args.steps, 8)), torch)
        kname = handle.kernel_name(True)
        ups = units / (ms * 1e-3)  # frame-iterations/s (ADMM) or frames/s (ML) of that kernel
        roof = fused_roofline(kname, ups, cus) or dict(bound="valu", frac=None, achieved=None, peak=round(cus * 4 * NOMINAL_CLOCK_HZ / 1e9, 1),
                                                        unit="G VALU issue cycles/s (all SIMDs)", note="no committed PMC counters for %s: run tools/collect_rooflines.sh" % kname)
        gbs = ups * bytes_per_frame_iter / 1e9
        roof.update(kernel=kname, kernel_class="admm_decode" if alg == "ADMM" else "ml_simulate", avg_launch_ms=round(ms, 4), traffic=None,
                    units_per_s=round(ups, 1), unit_of_work="frame-iteration" if alg == "ADMM" else "frame",
                    hbm_model={"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "hbm_model_frac": round(gbs / HBM_PEAK_GBS, 4),
                               "flag": "on-chip kernel: the state never leaves the CU, this bounds nothing" if handle.last_stats()[0] != "admm-stream" else "streaming kernels",
                               "note": "%s x %d B / HIP-event time of the kernel" % ("frame-iterations" if alg == "ADMM" else "frames", bytes_per_frame_iter)})
    decoder_name = {"MSA": "min-sum", "SPA": "sum-product", "BEC": "erasure decoder", "ADMM": "ADMM LP decoder (mu 3, eps 1e-5)", "ML": "ML (exhaustive search)"}[alg]
    channel_name = {"biawgn": "BI-AWGN", "bsc": "BSC", "bec": "BEC"}[channel]
    out = {
        "metric": ("decoded frames/s, n=1200 (3,6) min-sum max_iter=50 (+ roofline of the dominant kernel)"
                   if args.code == "1200_3_6_rand_ldpc_1" and args.max_iter == 50 and msa_biawgn else
                   "decoded frames/s, %s %s over %s max_iter=%d (+ roofline of the dominant kernel)" % (args.code, decoder_name, channel_name, args.max_iter)),
        "value": head["frames_per_s"], "unit": "frames/s", "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "ms_per_step_min": head["ms_per_step_min"], "ms_per_step_max": head["ms_per_step_max"],
        "timed_blocks": head["timed_blocks"], "blocks_ms_per_step": [round(1e3 * b / args.steps, 4) for b in res["blocks"]],
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u64 bit planes (2-bit messages, 32 frames per word pair)" if alg == "BEC" else "f32 arithmetic, f16 message storage" if f16 else precision,
        "data": "synthetic",
        "config": {"workload": "%s %s (%s) over %s, max_iter=%d, %s, %s (mean %.2f sweeps/frame), all-zero word + Philox noise on device" % (
                       args.code, args.decoder if alg != "BEC" else "%s = the ternary erasure decoder of the bec selector" % args.decoder, decoder_name,
                       channel_name, args.max_iter,
                       ("batch=%d frames per step over all GPUs" % per_round) if strong else ("batch=%d frames/GPU" % args.batch),
                       param_label(channel, param), head["mean_sweeps"]),
                   "code": args.code, "n": code.n, "m": code.m, "E": int(code.E), "decoder": args.decoder if alg != "BEC" else alg, "channel": channel,
                   ("snr_db" if channel == "biawgn" else "param"): param,
                   "max_iter": args.max_iter, "batch_per_gpu": rank_batch if strong else args.batch, "total_batch": per_round, "backend": backend_used,
                   "sweep_histogram_bins": hist_bins, "steps_per_launch": sim.rounds_per_launch() if hasattr(sim, "rounds_per_launch") else 1,
                   "parallelism": "frames sharded over %d GPU(s) by global frame index, 1 all-reduce of counters per step, %d step(s) in flight" % (
                       comm.world, sim.DEPTH)},
        "mean_sweeps": head["mean_sweeps"], "wer": head["wer"], "ber": head["ber"],
        "frames_counted": int(c[0]), "word_errors": int(c[1]), "bit_errors": int(c[2]),
        "kernel_ms_per_step": head.get("kernel_ms_per_step"), "host_overhead_ms_per_step": head.get("host_overhead_ms_per_step"),
        "side_kernels_ms_per_step": head.get("side_kernels_ms_per_step"),
        "algorithmic_GBps": head["algorithmic_GBps"], "bytes_per_frame_sweep": bytes_per_frame_iter,
        "roofline": roof,
        "collective": collective,
        "decode_from_hbm": hbm_leg,
        "fp32_mode": None,
        "points": [summarise(snr, r, max(4, args.steps)) for snr, r in extra],
    }
    if f32_res is not None:
        f32 = dict(summarise(param, f32_res, args.steps), backend=f32_res["backend"], kernel=f32_res["kernel"],
                   note="same workload with fp32 message arithmetic (bench.py --precision f32)")
        fp = f32_res["profile"]
        if fp and fp["fused_decode"][1] > 0:
            f32["roofline"] = fused_roofline(f32_res["kernel"], int(f32_res["counters"][3]) / (fp["fused_decode"][0] * 1e-3), cus)
        out["fp32_mode"] = f32
        if exact_res is not None and "error" not in exact_res:
            ec = exact_res["counters"]
            out["exact_fp32_mode"] = {
                "frames_per_s": round(int(ec[0]) / exact_res["seconds"], 1), "ms_per_step": round(1e3 * exact_res["seconds"] / exact_res["steps"], 4),
                "prior_grid": "2^-8", "frames": int(ec[0]), "frames_redone_in_fp64": int(exact_res["redone"]),
                "mean_sweeps": round(int(ec[3]) / max(int(ec[0]), 1), 3), "wer": round(int(ec[1]) / max(int(ec[0]), 1), 6),
                "note": "fp32 LDS kernel (k_fused_bp_grid) on LLRs rounded to multiples of 2^-8 with the in-kernel exactness guard; frames whose "
                        "messages leave the exact range are decoded again in fp64: decisions, iteration counts and counters equal the fp64 "
                        "reference's on the same priors for EVERY frame (tests/test_gpu_exact_fp32.py).  %d round(s) in flight.  "
                        "Not `value`: the priors differ from the unquantised workload's." % exact_res.get("depth", 1)}
        elif exact_res is not None:
            out["exact_fp32_mode"] = exact_res
    if f16_res is not None:
        fc, fp = f16_res["counters"], f16_res["profile"]
        b16 = 8 * code.E + 4 * code.n
        pair_ms = fp["stream_check_pass"][0] + fp["stream_variable_pass"][0]
        out["fp16_storage_mode"] = {
            "frames_per_s": round(int(fc[0]) / f16_res["seconds"], 1), "ms_per_step": round(1e3 * f16_res["seconds"] / args.steps, 4),
            "speedup_over_fp32": round(int(fc[0]) / f16_res["seconds"] / head["frames_per_s"], 3),
            "mean_sweeps": round(int(fc[3]) / max(int(fc[0]), 1), 3), "wer": round(int(fc[1]) / max(int(fc[0]), 1), 6),
            "bytes_per_frame_sweep": b16, "sweep_achieved_GBps": round(int(fc[3]) * b16 / (pair_ms * 1e-3) / 1e9, 1),
            "sweep_frac_of_hbm_peak": round(int(fc[3]) * b16 / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "check_pass_ms": round(fp["stream_check_pass"][0] / max(fp["stream_check_pass"][1], 1), 4),
            "variable_pass_ms": round(fp["stream_variable_pass"][0] / max(fp["stream_variable_pass"][1], 1), 4),
            "note": "LDPC_DTYPE_F16: check <-> variable messages stored as fp16 (two-array sweep: every E-sized line moved once per pass, "
                    "8E + 4n bytes per frame-sweep), arithmetic, priors and marginals fp32.  Tolerance mode: marginals within 1e-2 (1 + |fp32|) "
                    "over three sweeps, published WER points within 4 sigma, identical-noise WER equal to fp32's within counting noise "
                    "(tests/test_gpu_f16_storage.py).  Never `value`."}
    if stream_res is not None:
        sp, sc = stream_res["profile"], stream_res["counters"]
        it_sum = int(sc[3])
        legs = {}
        for kname, share in (("stream_check_pass", (2 * code.E + code.n) * s), ("stream_variable_pass", (code.E + 2 * code.n) * s)):
            kms, kl = sp[kname]
            if kl:
                gbs = it_sum * share / (kms * 1e-3) / 1e9
                legs[kname] = {"achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "avg_launch_ms": round(kms / kl, 4),
                               "launches": int(kl), "compulsory_bytes_per_frame_sweep": share}
        pair_ms = sp["stream_check_pass"][0] + sp["stream_variable_pass"][0]
        out["roofline_streaming_backend"] = {
            "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernels": legs,
            "sweep_achieved": round(it_sum * bytes_per_frame_iter / (pair_ms * 1e-3) / 1e9, 1),
            "sweep_frac": round(it_sum * bytes_per_frame_iter / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frames_per_s": round(int(sc[0]) / stream_res["seconds"], 1),
            "note": "same workload with --backend stream (state resident in HBM): the HBM-bound path used for codes that do not fit the LDS; "
                    "sweep_* = executed frame-sweeps x s(4E+n) / time of the two passes; per kernel: its own compulsory bytes"}
    out["cpu_baseline"] = cpu_base  # rank 0's host; None only with --no-cpu-baseline
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; ms_per_step is the median block")
    ap.add_argument("--batch", type=int, default=65536, help="frames per GPU per step (weak scaling: the whole job decodes N x this per step)")
    ap.add_argument("--total-batch", type=int, default=None,
                    help="frames per step over ALL GPUs (strong scaling: each rank decodes its shard; BASELINE configs 4 / 5 state 2^20 / 2^18 for 8 GPUs)")
    ap.add_argument("--decoder", default="MSA", choices=["MSA", "SPA", "BEC", "ADMM", "ML"],
                    help="decoder selector of the reference's CLI (src/main.py:12, src/utils.py:16); ADMM (src/admm.py:42-69, mu = 3, eps = 1e-5) and ML "
                         "(exhaustive search over the codebook of a short code, src/biawgn.py:66-78) are the SURVEY 8(f) decoders")
    ap.add_argument("--channel", default="biawgn", choices=["biawgn", "bsc", "bec"],
                    help="channel selector (src/models.py:3); `bec` pairs with the ternary erasure decoder whatever --decoder says, as in the registry")
    ap.add_argument("--param", type=float, default=None, help="channel parameter: SNR in dB / crossover probability / erasure probability (default: --snr, 0.07, 0.40)")
    ap.add_argument("--snr", type=float, default=1.0)
    ap.add_argument("--max-iter", type=int, default=50)
    ap.add_argument("--code", default="1200_3_6_rand_ldpc_1")
    ap.add_argument("--precision", default="f64", choices=["f32", "f64", "f16"],
                    help="message arithmetic; f64 is the reference's own (hard decisions bit-identical to it), f32 the throughput mode, f16 = fp16 "
                         "STORAGE of the streaming messages with fp32 arithmetic (codes whose state lives in HBM; a tolerance mode)")
    ap.add_argument("--backend", default="auto", choices=["auto", "stream", "fused"])
    ap.add_argument("--points", type=float, nargs="*", default=None, help="extra channel parameters reported under 'points' (default: 2.0 3.0 dB / p = 0.05 / eps = 0.35)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0, help="CPU work per baseline leg (C port, scipy processes)")
    ap.add_argument("--no-profile", action="store_true", help="headline only: skip the HIP-event kernel pass and the side legs")
    ap.add_argument("--no-baseline-configs", action="store_true",
                    help="default workload only: skip the `baseline_configs` block (BASELINE configs 3-5, a few steps each, on the same line)")
    return ap.parse_args(argv)


# BASELINE.json configs 3-5 on the line of the DEFAULT command (`baseline_configs`): the selectors of src/main.py:11-12 / src/models.py:3 that
# the driver's command never passes.  Each entry = the flags of that configuration's own bench line (tools/round_measure.sh), a few steps.
BASELINE_CONFIGS = [
    ("config3_spa_bsc", "BASELINE configs[2]: sum-product over BSC p = 0.07, n = 1200 (3,6), 65 536 frames",
     ["--decoder", "SPA", "--channel", "bsc", "--param", "0.07", "--precision", "f32", "--batch", "65536", "--steps", "20", "--warmup", "3"]),
    ("config3_bec", "BASELINE configs[2]: the bec selector's ternary erasure decoder over BEC eps = 0.40, n = 1200 (3,6), 65 536 frames",
     ["--decoder", "SPA", "--channel", "bec", "--param", "0.40", "--batch", "65536", "--steps", "64", "--warmup", "8"]),
    ("config4", "BASELINE configs[3]: rate-1/2 irregular n = 10 000 min-sum over BI-AWGN, 131 072 frames per GPU (2^20 over 8 GPUs)",
     ["--code", "gen:irg:10000", "--batch", "131072", "--snr", "1.2", "--precision", "f32", "--steps", "3", "--warmup", "1"]),
    ("config5", "BASELINE configs[4]: (3,6)-regular n = 64 800 min-sum over BI-AWGN with per-frame early termination, 32 768 frames per GPU (2^18 over 8 GPUs)",
     ["--code", "gen:reg:64800:3:6", "--batch", "32768", "--snr", "2.0", "--precision", "f32", "--steps", "2", "--warmup", "1"]),
]


def baseline_config_args(args, flags):
    """argparse namespace of one `baseline_configs` entry: the entry's own flags, headline-only legs, this run's --gpus."""
    sub = parse_args(flags + ["--gpus", str(args.gpus), "--repeats", "3", "--points", "--max-iter", str(args.max_iter)])
    sub.headline_only = True
    return sub


def is_default_workload(args):
    """True for the command the round driver runs (the headline of BASELINE.json, no selector passed): only that line carries `baseline_configs`."""
    d = parse_args([])
    return all(getattr(args, k) == getattr(d, k) for k in ("decoder", "channel", "param", "snr", "max_iter", "code", "precision", "backend", "batch",
                                                            "total_batch")) and not args.no_profile and not args.no_baseline_configs


def condense(line):
    """What a `baseline_configs` entry keeps of a full bench line."""
    keep = ("value", "unit", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "timed_blocks", "steps", "warmup", "n_gpus", "scaling", "dtype",
            "mean_sweeps", "wer", "ber", "frames_counted", "kernel_ms_per_step", "side_kernels_ms_per_step", "host_overhead_ms_per_step",
            "algorithmic_GBps", "bytes_per_frame_sweep", "roofline", "cpu_baseline")
    out = {"frames_per_s": line["value"], "workload": line["config"]["workload"], "backend": line["config"]["backend"],
           "steps_per_launch": line["config"]["steps_per_launch"]}
    out.update({k: line.get(k) for k in keep})
    return out


def self_launch(args, argv, needs_gpus=True):
    """`python bench.py --gpus N` (N > 1) started as a PLAIN process: this parent -- which has made no GPU call and makes none -- starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... <this script> <same argv>` as a fresh child process (never exec: a
    process must not be replaced once anything may have touched the GPU), relays rank 0's JSON line and returns the child's exit code.
    With fewer than N GPUs: one line saying so, exit code 3 -- never an `n_gpus: 1` line for a `--gpus N` request."""
    import socket
    import subprocess

    if needs_gpus:
        import torch  # device_count() enumerates without initialising the runtime

        have = torch.cuda.device_count()
        # LDPC_DIST_BACKEND=gloo (tests/test_gpu_multirank.py): several ranks may share a GPU -- RCCL needs one GPU per rank
        if have < (1 if os.environ.get("LDPC_DIST_BACKEND") == "gloo" else args.gpus):
            print("bench.py: --gpus %d requested but this node has %d GPU(s); no line printed" % (args.gpus, have), file=sys.stderr)
            return 3
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0])] + list(argv)
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in child.stdout:  # rank 0's JSON line goes to stdout untouched; anything else the launcher or a rank printed goes to stderr
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln)
        sys.stdout.flush()
    return child.wait()


def main(argv=None, make_handle=None, device="cuda"):
    """`make_handle` / `device` exist for tests/test_dist_cpu.py (a CPU stand-in for the decoder handle on gloo ranks, entered through a
    wrapper script so that the self-launch path is the one under test); bench.py itself always runs the HIP handle on a GPU."""
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, argv, needs_gpus=(device == "cuda")))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; refusing to print a line for another N" % (args.gpus, world_env), file=sys.stderr)
        sys.exit(3)
    # rank 0 times the CPU baselines first: nothing of HIP / torch.cuda / the process group exists in this process yet, so the scipy
    # leg may fork its workers, and at N > 1 the other ranks simply wait at the rendezvous meanwhile (~25 s headline + ~4 s per
    # baseline_configs entry at N = 1; the process-group timeout is 10 min)
    rank0 = int(os.environ.get("RANK", "0")) == 0
    subs = [(name, what, baseline_config_args(args, flags)) for name, what, flags in BASELINE_CONFIGS] if is_default_workload(args) else []
    cpu_base, sub_base = None, {}
    if rank0 and not args.no_cpu_baseline:
        alg, channel, param = resolve_workload(args.decoder, args.channel, args.param, args.snr)
        cpu_base = cpu_baseline(load_code(args.code), alg, channel, param, args.max_iter, args.precision, args.cpu_baseline_seconds)
        if world_env == 1:  # the contract: the CPU baseline is timed on rank 0 at N = 1 only; the sub-configurations take the C port alone
            for name, _, sub in subs:
                alg, channel, param = resolve_workload(sub.decoder, sub.channel, sub.param, sub.snr)
                sub_base[name] = cpu_baseline(load_code(sub.code), alg, channel, param, sub.max_iter, sub.precision,
                                              min(3.0, args.cpu_baseline_seconds), scipy_leg=False)
    from ldpc_decoders_amd import dist

    comm = dist.init_from_env(prefer_gpu=(device == "cuda"))
    try:
        out = run_bench(args, comm, make_handle=make_handle, device=device, cpu_base=cpu_base)
        block = {}
        for name, what, sub in subs:
            t0 = time.time()
            try:  # a sub-configuration must not cost the headline line -- but every rank takes the same branch (collectives inside)
                line = run_bench(sub, comm, make_handle=make_handle, device=device, cpu_base=sub_base.get(name))
                if comm.is_root:
                    block[name] = dict(condense(line), what=what, flags=" ".join(dict((n, f) for n, _, f in BASELINE_CONFIGS)[name]),
                                       wall_s=round(time.time() - t0, 1))
            except Exception as e:
                block[name] = {"error": repr(e), "what": what}
        if comm.is_root:
            if subs:
                out["baseline_configs"] = block
            print(json.dumps(out))
    finally:
        dist.finalize()


if __name__ == "__main__":
    main()
