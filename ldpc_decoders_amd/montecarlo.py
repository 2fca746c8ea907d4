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
    Returns dict(tot, wec, bec, iter_sum)."""
    n = len(x)
    tot = wec = bec = itsum = 0
    while wec < min_wec:
        if pick_word is not None:  # --codeword -1: a fresh random codeword per frame (src/main.py:38)
            chunk_now = 1
            x = pick_word()
        else:
            chunk_now = chunk
        X = np.broadcast_to(x, (chunk_now, n))
        Y = channel.send(X)
        xhat, iters = decoder.decode_batch(Y)
        err = (np.asarray(xhat) != X).sum(axis=1)
        for e, it in zip(err, iters):
            tot += 1
            wec += int(e > 0)
            bec += int(e)
            itsum += int(it)
            if wec >= min_wec:
                break
        if on_progress:
            on_progress(tot, wec, bec)
    return dict(tot=tot, wec=wec, bec=bec, iter_sum=itsum)


class DeviceSimulator:
    """Rounds of ``batch`` frames per rank, all on the GPU (channel kernel -> decode -> count)."""

    def __init__(self, handle, channel, max_iter, codeword=0, seed=0x5EED1200, comm=None, hist_bins=0):
        import torch

        self.torch = torch
        self.h, self.channel, self.max_iter, self.codeword = handle, channel, int(max_iter), int(codeword)
        self.seed, self.comm, self.hist_bins = int(seed), comm or Comm(), int(hist_bins)
        self.counters = torch.zeros(4 + self.hist_bins, dtype=torch.int64, device="cuda")

    def run_round(self, param, stream_id, frame0, frames_total, flags=0):
        """Decode global frames [frame0, frame0+frames_total) split over ranks; returns the reduced counters (numpy)."""
        start, cnt = self.comm.shard(frame0, frames_total)
        self.counters.zero_()
        if cnt > 0:
            self.h.simulate(self.channel, param, self.codeword, self.seed, stream_id, start, cnt, self.max_iter, self.counters,
                            flags=flags, hist_bins=self.hist_bins)
        red = self.comm.all_reduce_sum(self.counters)
        return red.cpu().numpy()

    def run_point(self, param, stream_id, min_wec, batch_per_rank, on_progress=None, max_frames=None):
        tot = np.zeros(4 + self.hist_bins, dtype=np.int64)
        frame0 = 0
        per_round = int(batch_per_rank) * self.comm.world
        while tot[_lib.CNT_WEC] < min_wec and (max_frames is None or tot[_lib.CNT_TOT] < max_frames):
            tot += self.run_round(param, stream_id, frame0, per_round)
            frame0 += per_round
            if on_progress:
                on_progress(int(tot[0]), int(tot[1]), int(tot[2]))
        out = dict(tot=int(tot[0]), wec=int(tot[1]), bec=int(tot[2]), iter_sum=int(tot[3]))
        if self.hist_bins:
            out["hist"] = tot[4:].tolist()
        return out
