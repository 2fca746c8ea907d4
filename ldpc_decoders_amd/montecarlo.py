"""Monte-Carlo error counting -- the build's counterpart of the loop in the reference's ``src/main.py:22-50``.

Two modes:
  * ``run_point_exact``  host numpy noise from the global ``np.random`` stream, frames decoded in chunks on the GPU,
    counters truncated at the first prefix reaching ``min_wec``: for a given ``np.random.seed`` this reproduces the
    reference's ``tot / wec / bec`` exactly (fp64 min-sum; SPA up to libm rounding).
  * ``run_point_device``  Philox noise generated on the GPU, keyed by the global frame index, so the counters do not
    depend on how the frame range is sharded over ranks; the stopping rule ``wec >= min_wec`` is evaluated once per
    round after ONE all-reduce of the counters (round-granular overshoot of ``tot`` vs the reference -- documented).
"""
import numpy as np

from . import _lib
from .dist import Comm


def run_point_exact(channel, decoder, x, min_wec, chunk=64, on_progress=None, pick_word=None):
    """Sequential-rule Monte-Carlo with host noise.  ``channel.send`` / ``decoder.decode_batch`` as in the registry.
    Returns dict(tot, wec, bec, iter_sum).

    The reference draws exactly one frame of noise per trip of its loop (src/main.py:37-40), so when it stops, the global
    ``np.random`` stream stands right behind the LAST COUNTED frame -- and the next ``--params`` value continues from there.
    Frames are decoded ``chunk`` at a time here; a chunk that reaches ``min_wec`` before its end is therefore rewound: the
    stream state saved before the chunk is restored and exactly the counted frames are drawn again."""
    n = len(x)
    tot = wec = bec = itsum = 0
    while wec < min_wec:
        if pick_word is not None:  # --codeword -1: a fresh random codeword per frame (src/main.py:38)
            chunk_now = 1
            x = pick_word()
        else:
            chunk_now = chunk
        X = np.broadcast_to(x, (chunk_now, n))
        state = np.random.get_state() if chunk_now > 1 else None
        Y = channel.send(X)
        xhat, iters = decoder.decode_batch(Y)
        err = (np.asarray(xhat) != X).sum(axis=1)
        used = 0
        for e, it in zip(err, iters):
            used += 1
            tot += 1
            wec += int(e > 0)
            bec += int(e)
            itsum += int(it)
            if wec >= min_wec:
                break
        if used < chunk_now:  # stopped inside the chunk: leave the stream where the reference leaves it
            np.random.set_state(state)
            channel.send(np.broadcast_to(x, (used, n)))
        if on_progress:
            on_progress(tot, wec, bec)
    return dict(tot=tot, wec=wec, bec=bec, iter_sum=itsum)


class DeviceSimulator:
    """Rounds of ``batch`` frames per rank, all on the GPU (channel kernel -> decode -> count).

    Rounds are PIPELINED: ``launch_round`` only enqueues work (zero the round's counters, the simulate kernels, the all-reduce
    over ranks, a copy of the reduced counters into page-locked host memory, an event) and returns at once; ``finish_round`` waits
    for that round's event alone.  With two rounds in flight the GPU never idles behind the host, the collective of round k
    overlaps the decode of round k+1, and no host synchronisation sits between kernels.  ``run_round`` = launch + finish (the
    synchronous form)."""

    DEPTH = 2  # rounds (or blocks of rounds, launch_rounds) in flight in run_point / bench.py

    def __init__(self, handle, channel, max_iter, codeword=0, seed=0x5EED1200, comm=None, hist_bins=0, device="cuda", prior_grid=None):
        """``device``: where the counters live -- "cuda" always in the product (the handle is a HIP decoder); "cpu" lets the N > 1 driver
        layer (sharding, pipelining, collective, stopping rule) be exercised on gloo ranks with a stand-in handle (tests/test_dist_cpu.py)."""
        import torch

        self.torch = torch
        self.h, self.channel, self.max_iter, self.codeword = handle, channel, int(max_iter), int(codeword)
        self.seed, self.comm, self.hist_bins = int(seed), comm or Comm(), int(hist_bins)
        self.device = device
        # exact-in-fp32 mode (include/ldpc_hip.h LDPC_FLAG_PRIOR_GRID): priors on a 2^-k grid, fp32 kernels under the exactness guard, the
        # frames beyond it redone in fp64 by the handle; `redone` counts them
        self.prior_grid, self.redone = prior_grid, 0
        if prior_grid is not None and (channel != "biawgn" or int(codeword) not in (0, 1)):
            raise ValueError("--prior-grid: min-sum over BI-AWGN with the all-zero / all-one word")
        # exact-in-fp32 rounds carry two more words behind the counters: frames re-decoded in fp64, and a flag "the redo list overflowed"
        k = 4 + self.hist_bins + (2 if prior_grid is not None else 0)
        on_gpu = device == "cuda"
        self._slots = [dict(dev=torch.zeros(k, dtype=torch.int64, device=device),
                            host=torch.zeros(k, dtype=torch.int64).pin_memory() if on_gpu else torch.zeros(k, dtype=torch.int64),
                            done=torch.cuda.Event() if on_gpu else None, busy=False) for _ in range(self.DEPTH + 1)]

        self._next = 0
        # (A two-stream variant -- consecutive rounds on two HIP streams with two decoders -- was measured in round 4: +0.35 % / +0.85 % on the
        # 5.3 / 2.75 ms rounds of the n = 1200 kernels, -15 % on the erasure decoder's 0.25 ms rounds; removed in round 6, HISTORY.md.)
        self._multi = []  # slots of launch_rounds: [rounds, k] counter blocks

    def launch_round(self, param, stream_id, frame0, frames_total, flags=0):
        """Enqueue the decode of global frames [frame0, frame0+frames_total), split over ranks; returns a ticket for finish_round."""
        slot = self._slots[self._next]
        if slot["busy"]:
            raise RuntimeError("more than %d rounds in flight" % len(self._slots))
        self._next = (self._next + 1) % len(self._slots)
        start, cnt = self.comm.shard(frame0, frames_total)
        if self.prior_grid is not None:
            return self._launch_exact_round(slot, param, stream_id, start, cnt)
        slot["dev"].zero_()
        if cnt > 0:
            self.h.simulate(self.channel, param, self.codeword, self.seed, stream_id, start, cnt, self.max_iter, slot["dev"],
                            flags=flags, hist_bins=self.hist_bins)
        self.comm.all_reduce_sum(slot["dev"], async_on_stream=True)
        slot["host"].copy_(slot["dev"], non_blocking=True)
        if slot["done"] is not None:
            slot["done"].record()
        slot["busy"] = True
        return slot

    def rounds_per_launch(self):
        """Rounds worth sending in ONE launch (``launch_rounds``): > 1 only where the kernel keeps its frame positions busy across round
        boundaries (the LDS-resident erasure decoder).  Every round keeps its own counter row, so counters stay a function of the round
        size alone."""
        h = self.h
        if self.codeword == -1 or self.device != "cuda" or not hasattr(h, "rounds_per_launch"):
            return 1
        if self.prior_grid is not None:  # exact-in-fp32 mode: one fp64 redo pass per block of eight guarded launches
            return 8 if h.last_stats()[0] == "fused" else 1
        return int(h.rounds_per_launch())

    def launch_rounds(self, param, stream_id, frame0, frames_total, rounds, flags=0):
        """Enqueue ``rounds`` consecutive rounds of ``frames_total`` global frames each -- round r covers [frame0 + r * frames_total, + frames_total),
        split over ranks like ``launch_round`` -- as one call with one counter row per round and ONE all-reduce of the whole block; returns a
        ticket for ``finish_rounds``."""
        torch = self.torch
        k = 4 + self.hist_bins + (2 if self.prior_grid is not None else 0)
        blk = None
        for b in self._multi:
            if not b["busy"] and b["dev"].shape[0] == rounds:
                blk = b
        if blk is None:
            blk = dict(dev=torch.zeros((rounds, k), dtype=torch.int64, device=self.device),
                       host=torch.zeros((rounds, k), dtype=torch.int64).pin_memory() if self.device == "cuda" else torch.zeros((rounds, k), dtype=torch.int64),
                       done=torch.cuda.Event() if self.device == "cuda" else None, busy=False)
            self._multi.append(blk)
        start, cnt = self.comm.shard(frame0, frames_total)
        blk["dev"].zero_()
        if cnt > 0 and self.prior_grid is not None:
            for r in range(rounds):
                self.h.simulate(self.channel, param, self.codeword, self.seed, stream_id, start + r * frames_total, cnt, self.max_iter, blk["dev"][r],
                                flags=_lib.flag_prior_grid(self.prior_grid), hist_bins=self.hist_bins)
            self.h.redo_on_stream(param, self.codeword, self.seed, stream_id, self.max_iter, self.prior_grid, blk["dev"], blk["dev"][0][4 + self.hist_bins:],
                                  start, frames_total, hist_bins=self.hist_bins)
        elif cnt > 0:
            self.h.simulate_rounds(self.channel, param, self.codeword, self.seed, stream_id, start, cnt, rounds, frames_total, self.max_iter,
                                   blk["dev"], flags=flags, hist_bins=self.hist_bins)
        self.comm.all_reduce_sum(blk["dev"], async_on_stream=True)
        blk["host"].copy_(blk["dev"], non_blocking=True)
        if blk["done"] is not None:
            blk["done"].record()
        blk["busy"] = True
        return blk

    def finish_rounds(self, blk, counted=True):
        """Wait for one ``launch_rounds`` ticket; returns its whole-job counter rows (numpy int64 [rounds, 4 + hist_bins]).
        ``counted=False``: a speculative block that the stopping rule discards -- its fp64 redo frames do not enter ``redone``."""
        if blk["done"] is not None:
            blk["done"].synchronize()
        blk["busy"] = False
        out = blk["host"].numpy().copy()
        if self.prior_grid is not None:
            k = 4 + self.hist_bins
            self._check_redo(out[0, k:], counted)
            out = out[:, :k]
        return out

    # ---- exact-in-fp32 mode: the rounds stay in flight.  The guarded fp32 kernels run back to back; the frames they set aside stay on a
    # device-resident list and are re-decoded in fp64 by ONE redo pass per block of rounds, enqueued behind the block's last kernel on the
    # same stream -- priors drawn again on the device from the listed frame indices, every frame counted into the row of its own round --
    # followed by the all-reduce and the copy-out.  No host round trip anywhere; the redo pass (0.25 ms: a few frames, fifty sweeps of
    # latency) is paid once per block instead of once per round.
    def _launch_exact_round(self, slot, param, stream_id, start, cnt):
        k = 4 + self.hist_bins
        slot["dev"].zero_()
        if cnt > 0:
            self.h.simulate(self.channel, param, self.codeword, self.seed, stream_id, start, cnt, self.max_iter, slot["dev"],
                            flags=_lib.flag_prior_grid(self.prior_grid), hist_bins=self.hist_bins)
            self.h.redo_on_stream(param, self.codeword, self.seed, stream_id, self.max_iter, self.prior_grid, slot["dev"], slot["dev"][k:],
                                  start, 0, hist_bins=self.hist_bins)
        self.comm.all_reduce_sum(slot["dev"], async_on_stream=True)
        slot["host"].copy_(slot["dev"], non_blocking=True)
        slot["done"].record()
        slot["busy"] = True
        return slot

    def _check_redo(self, extra, counted=True):
        """extra = [frames re-decoded in fp64, redo-list overflows] of a finished round / block."""
        if not counted:
            return
        if extra[1] != 0:
            raise _lib.LdpcHipError("prior grid 2^-%d: more than %d frames of one block of rounds beyond the exactness guard -- the grid is too fine "
                                    "for this operating point" % (self.prior_grid, self.h.REDO_ROWS))
        self.redone += int(extra[0])

    def finish_round(self, slot, counted=True):
        """Wait for one launched round; returns its whole-job counters (numpy int64).  ``counted=False``: a speculative round that the
        stopping rule discards (its fp64 redo frames do not enter ``redone``)."""
        if slot["done"] is not None:
            slot["done"].synchronize()
        slot["busy"] = False
        out = slot["host"].numpy().copy()
        if self.prior_grid is not None:
            k = 4 + self.hist_bins
            self._check_redo(out[k:], counted)
            out = out[:k]
        return out

    def run_round(self, param, stream_id, frame0, frames_total, flags=0):
        """Decode global frames [frame0, frame0+frames_total) split over ranks; returns the reduced counters (numpy)."""
        return self.finish_round(self.launch_round(param, stream_id, frame0, frames_total, flags))

    def pipeline_depth(self):
        """Rounds worth keeping in flight.  Only the fused in-kernel path returns from ``simulate`` without a host wait; the
        streaming kernels, the ADMM composition and ``--codeword -1`` poll / synchronise inside ``simulate``, so a second round in
        flight buys them nothing and would only be decoded for the bin when the stopping rule fires."""
        h = self.h
        if self.codeword == -1 or not hasattr(h, "last_stats"):
            return 1
        return self.DEPTH if h.last_stats()[0] == "fused" else 1

    def run_point(self, param, stream_id, min_wec, batch_per_rank, on_progress=None, max_frames=None):
        """Rounds until ``wec >= min_wec``.  The FIRST round runs on its own: most points of the experiment tables reach ``min_wec``
        (or ``max_frames``) inside it, and a speculative second round would be decoded and thrown away.  From the second round on
        (fused path) rounds are pipelined: the stopping rule is evaluated when a round's counters arrive, by which time the next
        round is already running; that round is drained and DISCARDED, so the counters are exactly those of the synchronous loop
        (a function of the round size only, not of the pipeline depth or the number of ranks).  ``on_progress(tot, wec, bec, hist)``
        receives the reduced counters after every counted round (``hist`` = the histogram bins or None)."""
        tot = np.zeros(4 + self.hist_bins, dtype=np.int64)
        frame0 = 0
        per_round = int(batch_per_rank) * self.comm.world
        inflight = []
        capped = False

        def more():
            return tot[_lib.CNT_WEC] < min_wec and (max_frames is None or tot[_lib.CNT_TOT] < max_frames)

        def count(got):
            nonlocal tot
            tot += got
            if on_progress:
                on_progress(int(tot[0]), int(tot[1]), int(tot[2]), tot[4:].copy() if self.hist_bins else None)

        count(self.run_round(param, stream_id, frame0, per_round))
        frame0 += per_round
        depth = self.pipeline_depth()
        rpl = self.rounds_per_launch()
        while rpl > 1 and (more() or inflight):
            # several rounds per launch: the rows are counted IN ORDER until the stopping rule fires; rows behind it (and a block already in
            # flight) are decoded and discarded, exactly like the speculative round of the pipelined loop below
            while more() and len(inflight) < depth:
                inflight.append(self.launch_rounds(param, stream_id, frame0, per_round, rpl))
                frame0 += per_round * rpl
            for row in self.finish_rounds(inflight.pop(0), counted=more()):  # (the redo count is per block: in, if its first row is)
                if more():
                    count(row)
        while more() or inflight:
            while more() and len(inflight) < depth:
                inflight.append(self.launch_round(param, stream_id, frame0, per_round))
                frame0 += per_round
            keep = more()
            got = self.finish_round(inflight.pop(0), counted=keep)
            if keep:
                count(got)
        if tot[_lib.CNT_WEC] < min_wec:
            capped = True  # stopped by max_frames, not by the word-error target
        out = dict(tot=int(tot[0]), wec=int(tot[1]), bec=int(tot[2]), iter_sum=int(tot[3]), capped=capped)
        if self.hist_bins:
            out["hist"] = tot[4:].tolist()
        return out
