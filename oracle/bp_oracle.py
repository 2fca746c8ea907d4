"""CPU oracle: numpy restatement of the reference's BP hot path (fp64, batched over frames).

TEST INFRASTRUCTURE ONLY.  Nothing under ``ldpc_decoders_amd/`` may import this
module; it exists so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` can check / time the HIP path against an
independent statement of the same algorithm.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this module against
(i) the six known-answer tests the reference carries (``src/biawgn.py:85-92``,
``src/bsc.py:82-89``, ``src/bec.py:132-139``) and (ii) golden vectors captured by
importing the reference in the build container (``oracle/make_goldens.py`` →
``tests/golden/*.npz``); ``tests/test_goldens_regenerate_cpu.py`` additionally re-runs
the live reference when ``/root/reference`` exists and compares with the committed files.

Everything is written over an explicit *edge list* in row-major order of H
(edge k = (chk[k], var[k]) sorted by check then variable), which is what the
reference gets from ``np.where(parity_mtx)`` (``src/bpa.py:12``, ``src/bec.py:77``).
Frames are batched on the leading axis; every frame follows exactly the
sequence of fp64 operations the reference applies to a single frame, and is
frozen at the iteration where the reference would have returned.
"""
import math

import numpy as np

MSA, SPA = "MSA", "SPA"


# --------------------------------------------------------------------------- graph
class Edges:
    """Row-major edge list of a parity-check matrix plus the per-rank index sets
    used to reproduce scipy's accumulation order.

    reference: ``BPA.__init__`` (src/bpa.py:9-15) builds ``xx, yy = np.where(H)`` and
    sums with ``coo_matrix(...).sum(axis)`` which accumulates, per output slot,
    from 0.0 in ascending storage (= edge) order.
    """

    def __init__(self, m, n, chk, var):
        chk = np.asarray(chk, dtype=np.int64)
        var = np.asarray(var, dtype=np.int64)
        order = np.lexsort((var, chk))
        assert (order == np.arange(len(chk))).all(), "edges must be row-major sorted"
        self.m, self.n, self.E = int(m), int(n), len(chk)
        self.chk, self.var = chk, var
        # rank of each edge inside its check (row) and inside its variable (column)
        self.row_rank = self._rank(chk)
        self.col_rank = self._rank(var)
        self.row_sets = [np.flatnonzero(self.row_rank == r) for r in range(self.row_rank.max() + 1)]
        self.col_sets = [np.flatnonzero(self.col_rank == r) for r in range(self.col_rank.max() + 1)]
        self.row_deg = np.bincount(chk, minlength=m)
        self.col_deg = np.bincount(var, minlength=n)

    @staticmethod
    def _rank(keys):
        # position of every edge among the edges sharing its key, in ascending edge order
        order = np.argsort(keys, kind="stable")
        sorted_keys = keys[order]
        start = np.r_[0, np.flatnonzero(np.diff(sorted_keys)) + 1]
        seg_len = np.diff(np.r_[start, len(keys)])
        pos = np.arange(len(keys)) - np.repeat(start, seg_len)
        rank = np.empty(len(keys), dtype=np.int64)
        rank[order] = pos
        return rank

    @classmethod
    def from_dense(cls, H):
        H = np.asarray(H)
        chk, var = np.nonzero(H)
        return cls(H.shape[0], H.shape[1], chk, var)

    def to_dense(self):
        H = np.zeros((self.m, self.n), dtype=np.int64)
        H[self.chk, self.var] = 1
        return H

    # ordered segmented sums: out[slot] = ((0 + d_a) + d_b) + ...
    def sum_rows(self, d):
        out = np.zeros(d.shape[:-1] + (self.m,), dtype=d.dtype)
        for idx in self.row_sets:
            out[..., self.chk[idx]] += d[..., idx]
        return out

    def sum_cols(self, d):
        out = np.zeros(d.shape[:-1] + (self.n,), dtype=d.dtype)
        for idx in self.col_sets:
            out[..., self.var[idx]] += d[..., idx]
        return out


def parse_parity_text(text):
    """Alist-like text -> Edges, with the reference loader's semantics.

    reference: ``codes.load_parity_mtx`` (src/codes.py:93-105): one check per
    non-blank line, whitespace separated variable numbers; base (0 or 1) is
    detected from the global minimum; n = max + (0 if base==1 else 1); every
    entry is stored at column ``var - 1`` *regardless of base*, so a 0-based
    file is cyclically shifted (variable 0 lands in the last column).
    """
    rows = [[int(t) for t in ln.split()] for ln in text.splitlines() if ln.split()]
    lo = min(min(r) for r in rows)
    hi = max(max(r) for r in rows)
    if lo not in (0, 1):
        raise Exception("Minimum index is not 0 or 1.")
    n = hi + (0 if lo == 1 else 1)
    chk, var = [], []
    for c, r in enumerate(rows):
        cols = sorted({(v - 1) % n for v in r})
        chk += [c] * len(cols)
        var += cols
    return Edges(len(rows), n, chk, var)


# --------------------------------------------------------------------------- check-node rules
def _sgn(v):
    # reference: mu.sign (src/math_utils.py:10): zero maps to +1 ; NaN maps to -1
    return np.where(v >= 0, 1.0, -1.0)


def _row_parity_sign(g, v):
    # reference: mu.prod_nonzero_sign (src/math_utils.py:38-43): (-1)^(#negatives in the row)
    neg = g.sum_rows((v < 0).astype(np.int64))
    return (neg % 2) * -2 + 1


def msa_check_update(g, v2c):
    """Plain min-sum check-node rule.  reference: ``MSA.decode_`` (src/bpa.py:86-102).

    extrinsic sign  = row sign parity / own sign (sgn(0)=+1);
    extrinsic |.|   = second minimum of the row at the FIRST arg-min edge, first
                      minimum everywhere else (``mu.csr_csc_argmax`` returns the first
                      maximiser in storage order, src/math_utils.py:78-94).
    """
    sign = _row_parity_sign(g, v2c)[..., g.chk] / _sgn(v2c)
    mag = np.abs(v2c)
    B = mag.shape[:-1]
    min1 = np.full(B + (g.m,), np.inf)
    for idx in g.row_sets:
        c = g.chk[idx]
        min1[..., c] = np.minimum(min1[..., c], mag[..., idx])
    # first arg-min position (rank) inside each row
    is_min = mag == min1[..., g.chk]
    arg1 = np.full(B + (g.m,), np.iinfo(np.int64).max, dtype=np.int64)
    for r, idx in enumerate(g.row_sets):
        c = g.chk[idx]
        cand = np.where(is_min[..., idx], r, np.iinfo(np.int64).max)
        arg1[..., c] = np.minimum(arg1[..., c], cand)
    at_arg1 = g.row_rank == arg1[..., g.chk]
    masked = np.where(at_arg1, np.inf, mag)
    min2 = np.full(B + (g.m,), np.inf)
    for idx in g.row_sets:
        c = g.chk[idx]
        min2[..., c] = np.minimum(min2[..., c], masked[..., idx])
    return sign * np.where(at_arg1, min2[..., g.chk], min1[..., g.chk])


def spa_check_update(g, v2c):
    """tanh-product check-node rule.  reference: ``SPA.decode_`` (src/bpa.py:71-75) with
    ``mu.prod_nonzero`` (src/math_utils.py:47-52: sign * exp(sum(log|t|))) and
    ``mu.arctanh`` (src/math_utils.py:56-60: exact +-1 -> +-inf).  The extrinsic
    value is obtained by DIVISION of the row product by the own factor.
    """
    with np.errstate(all="ignore"):
        t = np.tanh(v2c / 2.0)
        prod = _row_parity_sign(g, t) * np.exp(g.sum_rows(np.log(np.abs(t))))
        q = prod[..., g.chk] / t
        sat = np.abs(q) == 1
        out = np.where(sat, np.inf * q, np.arctanh(np.where(sat, 0.0, q)))
        return 2 * out


def spa_phi_check_update(g, v2c):
    """Sum-product check rule in the phi domain, leave-one-out, fp64 -- the statement the GPU's fp32 SPA mode follows:
    |c2v_j| = phi(sum_{i != j} phi(|v2c_i|)) with phi(x) = -log(tanh(x/2)) = log1p(2/expm1(x)); sign as in min-sum.

    Mathematically identical to ``spa_check_update`` (the reference formula).  It differs numerically only where the
    reference leaves ordinary arithmetic: tanh saturating to exactly +-1 in fp64 (|LLR| >~ 38 -> +-inf, atanh(>1) = NaN
    from exp(log(t))/t rounding, inf-inf = NaN -> marginal forced to 0) and v2c == 0 (0/0).  On frames whose messages
    stay below saturation the two produce the same decisions (tests/test_oracle_golden.py::test_phi_rule_*)."""
    with np.errstate(all="ignore"):
        a = np.abs(v2c)
        ph = np.log1p(2.0 / np.expm1(a))
        tot_pre = np.zeros(a.shape[:-1] + (g.m,))
        pre = np.empty_like(ph)
        for idx in g.row_sets:  # prefix sums in edge order
            c = g.chk[idx]
            pre[..., idx] = tot_pre[..., c]
            tot_pre[..., c] += ph[..., idx]
        tot_suf = np.zeros(a.shape[:-1] + (g.m,))
        mag = np.empty_like(ph)
        for idx in reversed(g.row_sets):  # suffix sums, last edge first
            c = g.chk[idx]
            s = pre[..., idx] + tot_suf[..., c]
            mag[..., idx] = np.log1p(2.0 / np.expm1(s))
            tot_suf[..., c] += ph[..., idx]
        flip = (_row_parity_sign(g, v2c)[..., g.chk] < 0) != ~(v2c >= 0)
        return np.where(flip, -mag, mag)


# --------------------------------------------------------------------------- flooding BP
def syndrome_ok(g, word):
    """reference: ``((H @ x_hat) % 2 == 0).all()`` (src/bpa.py:29); ``word`` may be the
    raw received vector at iteration 0 (real valued for BI-AWGN)."""
    s = g.sum_rows(np.asarray(word)[..., g.var])
    return (np.mod(s, 2) == 0).all(axis=-1)


def bp_decode(g, alg, y, priors, max_iter, return_trace=False):
    """Batched ``bpa.{SPA,MSA}(H, max_iter=..).decode(y, priors)``.

    reference: ``BPA.decode`` (src/bpa.py:17-63).  y, priors: [B, n] (or [n]).
    Returns (x_hat [B,n] float64 -- raw ``y`` rows where the frame left at
    iteration 0, else 0/1 -- , iters [B] = number of check/variable sweeps run).
    ``return_trace`` additionally returns the list of marginals per sweep
    (NaN->0 applied, as the reference leaves them) for frames still running.
    """
    cn = {MSA: msa_check_update, SPA: spa_check_update, "SPA_PHI": spa_phi_check_update}[alg]
    y = np.atleast_2d(np.asarray(y, dtype=np.float64))
    priors = np.atleast_2d(np.asarray(priors, dtype=np.float64))
    B = y.shape[0]
    x_hat = y.copy()
    iters = np.zeros(B, dtype=np.int64)
    live = np.ones(B, dtype=bool)
    v2c = priors[:, g.var].copy()
    trace = []
    bp_decode.last_peak = np.zeros(B)  # largest |v2c| each frame ever fed to a check (saturation diagnostics)
    it = 0
    while live.any():
        if 0 < max_iter <= it:
            break
        live &= ~syndrome_ok(g, x_hat)
        if not live.any():
            break
        L = np.flatnonzero(live)
        with np.errstate(all="ignore"):
            bp_decode.last_peak[L] = np.fmax(bp_decode.last_peak[L], np.abs(v2c[L]).max(axis=1))
            c2v = cn(g, v2c[L])
            marginal = priors[L] + g.sum_cols(c2v)
            v2c[L] = marginal[:, g.var] - c2v
        marginal[np.isnan(marginal)] = 0.0
        x_hat[L] = (marginal < 0).astype(np.float64)
        iters[L] += 1
        it += 1
        if return_trace:
            full = np.full((B, g.n), np.nan)
            full[L] = marginal
            trace.append(full)
    return (x_hat, iters, trace) if return_trace else (x_hat, iters)


# --------------------------------------------------------------------------- erasure decoder
def bec_decode(g, y, max_iter):
    """Batched ``bec.SPA(p, code, max_iter=..).decode(y)`` (== ``bec.MSA``).

    reference: src/bec.py:83-122.  Symbols {0,1,2(erased)}; messages {-1 (bit 0), +1 (bit 1), 0}.
    Returns (x_hat [B,n] int64 in {0,1,2}, iters [B]).
    """
    y = np.atleast_2d(np.asarray(y, dtype=np.int64))
    B = y.shape[0]
    msg_of = np.array([-1, 1, 0], dtype=np.int64)  # bec.py:76
    sym_of = np.array([2, 1, 0], dtype=np.int64)  # bec.py:75, indexed by sign in {0,1,-1}
    priors = msg_of[y]
    v2c = priors[:, g.var].copy()
    c2v = np.zeros_like(v2c)
    x_hat = y.copy()
    iters = np.zeros(B, dtype=np.int64)
    live = np.ones(B, dtype=bool)
    it = 0
    while live.any():
        if 0 < max_iter <= it:
            break
        live &= (x_hat == 2).any(axis=1)  # bec.py:97
        if not live.any():
            break
        L = np.flatnonzero(live)
        v, c = v2c[L], c2v[L]
        erased = g.sum_rows(1 - np.abs(v))[:, g.chk]  # bec.py:100
        ones_par = g.sum_rows((v > 0).astype(np.int64))[:, g.chk] % 2  # bec.py:110
        c = np.where(erased == 0, v, c)  # echo (bec.py:105)
        c = np.where(erased > 1, 0, c)
        c = np.where(erased == 1, (1 - np.abs(v)) * (2 * ones_par - 1), c)  # bec.py:108-112
        marginal = priors[L] + g.sum_cols(c)
        v2c[L] = np.sign(marginal[:, g.var] - c)
        c2v[L] = c
        x_new = sym_of[np.sign(marginal)]
        stop = (x_new == x_hat[L]).all(axis=1)  # stopping set (bec.py:120) -> return OLD x_hat
        keep = L[~stop]
        x_hat[keep] = x_new[~stop]
        iters[L] += 1  # sweeps executed (the stopping-set sweep included)
        live[L[stop]] = False
        it += 1
    return x_hat, iters


# --------------------------------------------------------------------------- channels / LLRs
def biawgn_noise_var(snr_db):
    return 10 ** (-snr_db / 10)  # src/biawgn.py:10


def biawgn_send(x, snr_db, rng=np.random):
    # src/biawgn.py:17-18 ; consumes the legacy global MT19937 stream exactly like the reference
    return (2 * x - 1) + rng.normal(0, np.sqrt(biawgn_noise_var(snr_db)), x.shape)


def biawgn_priors(y, snr_db):
    return -2 * y / biawgn_noise_var(snr_db)  # src/biawgn.py:28


def bsc_send(x, p, rng=np.random):
    return (x + (rng.random(x.shape) < p)) % 2  # src/bsc.py:16


def bsc_priors(y, p):
    return (np.log(1 - p) - np.log(p)) * (1 - 2 * y)  # src/bsc.py:21,25


def bec_send(x, p, rng=np.random):
    tt = (rng.random(x.shape) < p).astype(int)  # src/bec.py:17-18
    return np.clip(x + tt * 10, 0, 2)


def channel_decode(g, channel, alg, param, y, max_iter):
    """The registry-level call ``models[channel].<alg>(param, code, max_iter=..).decode(y)``."""
    if channel == "bec":
        return bec_decode(g, y, max_iter)
    pri = biawgn_priors(np.asarray(y, float), param) if channel == "biawgn" else bsc_priors(np.asarray(y), param)
    return bp_decode(g, alg, y, pri, max_iter)


# --------------------------------------------------------------------------- Monte-Carlo driver
def run_point(g, channel, alg, param, codeword, min_wec, max_iter, chunk=64, rng=np.random):
    """Counters of one ``--params`` point of the reference driver for the global numpy
    stream.  reference: ``main.test`` (src/main.py:22-50): frames are drawn and decoded
    one at a time until ``wec >= min_wec``; here they are drawn ``chunk`` at a time
    (``normal(size=(c,n))`` == c sequential ``size=n`` draws) and the counters are
    truncated at the first prefix reaching ``min_wec``.  NOTE: consumes up to
    chunk-1 frames of extra random numbers beyond the reference's stopping frame.
    """
    x = np.full(g.n, codeword, dtype=np.int64)
    send = {"biawgn": biawgn_send, "bsc": bsc_send, "bec": bec_send}[channel]
    tot = wec = bec = 0
    while wec < min_wec:
        Y = send(np.broadcast_to(x, (chunk, g.n)), param, rng)
        X, _ = channel_decode(g, channel, alg, param, Y, max_iter)
        err = (X != x).sum(axis=1)
        for e in err:
            tot += 1
            wec += int(e > 0)
            bec += int(e)
            if wec >= min_wec:
                break
    return tot, wec, bec


# --------------------------------------------------------------------------- counter-based RNG (device noise)
_PHILOX_M0, _PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32(counter, key, rounds=10):
    """Philox4x32-10 (Salmon et al., SC'11 'Parallel random numbers: as easy as 1, 2, 3').
    counter: uint32[...,4], key: uint32[...,2] -> uint32[...,4].  Vectorised; this is the
    integer stream the HIP channel kernels must reproduce bit-for-bit."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0 = np.uint64(_PHILOX_M0) * c[0]
        p1 = np.uint64(_PHILOX_M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(_PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(_PHILOX_W1)) & mask
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def philox_frame_words(seed, stream, frame, n_words):
    """uint32 words [n_words] for one frame: counter = (j, 0, frame_lo, frame_hi), key = (seed_lo ^ stream*..., seed_hi).
    Matches ``philox_word_block`` in ldpc_decoders_amd/csrc/ldpc_rng.hpp."""
    nblk = (n_words + 3) // 4
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nblk, dtype=np.uint32)
    ctr[:, 1] = np.uint32(stream & 0xFFFFFFFF)
    ctr[:, 2] = np.uint32(frame & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32((frame >> 32) & 0xFFFFFFFF)
    key = np.zeros((nblk, 2), dtype=np.uint32)
    key[:, 0] = np.uint32(seed & 0xFFFFFFFF)
    key[:, 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    return philox4x32(ctr, key).reshape(-1)[:n_words]


def u01_from_u32(w):
    """(w + 0.5) * 2^-32 in fp32 arithmetic semantics used on device: open interval (0,1)."""
    return (w.astype(np.float64) + 0.5) * (1.0 / 4294967296.0)


def device_biawgn_noise(seed, stream, frame, n):
    """fp64 model of the device Box-Muller: words (2i, 2i+1) -> normals (2i, 2i+1)."""
    npairs = (n + 1) // 2
    w = philox_frame_words(seed, stream, frame, 2 * npairs)
    u1, u2 = u01_from_u32(w[0::2]), u01_from_u32(w[1::2])
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.empty(2 * npairs)
    z[0::2] = r * np.cos(2 * math.pi * u2)
    z[1::2] = r * np.sin(2 * math.pi * u2)
    return z[:n]
