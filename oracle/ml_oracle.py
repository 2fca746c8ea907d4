"""CPU restatement of the reference's maximum-likelihood decoders (codebook search).

TEST INFRASTRUCTURE -- imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package never imports it.  Pinned by tests/golden/ml_vectors.npz (log-likelihood matrices and picks captured from the
reference, oracle/make_goldens_ml.py).

    BI-AWGN  reference src/biawgn.py:66-78
    BSC      reference src/bsc.py:63-75
    BEC      reference src/bec.py:21-36
    pick     reference src/math_utils.py:72-74 (arg_max_rand: np.random.choice among the maximisers)
"""
import numpy as np


def ml_coefficients(channel, param):
    """The constants the reference's constructors compute, as (c0, c1)."""
    if channel == "biawgn":
        return 2 * (10 ** (-param / 10)), 0.0  # src/biawgn.py:10,75
    with np.errstate(divide="ignore"):
        return np.log(param), np.log(1 - param)  # src/bsc.py:67 / src/bec.py:25


def ml_log_prob(channel, cb, y, coef):
    """log-likelihood of every codeword for one received word y [n] -> float64 [K]; numpy does the arithmetic exactly as
    upstream (same expressions, hence the same summation order)."""
    cb = np.asarray(cb)
    y = np.asarray(y)
    if channel == "biawgn":
        exponent = -np.square(cb * 2 - 1 - y) / coef[0]  # src/biawgn.py:75 ; coef[0] = 2 * noise_var
        return np.sum(exponent, axis=1)
    log_p, log_1p = coef
    if channel == "bsc":
        num_agrees = np.sum(cb == y, axis=1)  # src/bsc.py:71-73
        num_diffs = cb.shape[1] - num_agrees
        return num_diffs * log_p + num_agrees * log_1p
    num_erasures = np.sum(y > 1)  # src/bec.py:29-34
    num_agrees = np.sum(cb == y, axis=1)
    num_diffs = cb.shape[1] - num_agrees - num_erasures
    with np.errstate(invalid="ignore"):
        log_prob = num_erasures * log_p + num_agrees * log_1p
    log_prob = np.asarray(log_prob, dtype=np.float64)
    log_prob[num_diffs > 0] = -np.inf
    return log_prob


def ml_tie_set(log_prob):
    return np.argwhere(log_prob == np.max(log_prob)).flatten()  # src/math_utils.py:73


def ml_decode(channel, cb, y, coef):
    """One frame, pick from numpy's global stream exactly like upstream."""
    ind = np.random.choice(ml_tie_set(ml_log_prob(channel, cb, y, coef)), 1)[0]  # src/math_utils.py:74
    return np.asarray(cb)[ind]


def numpy_pairwise_sum(a):
    """Explicit statement of the order in which np.sum(x, axis=1) adds a contiguous row of n <= 128 doubles (numpy's
    pairwise_sum + the reduction's 0.0 start) -- what the device kernel implements; checked against np.sum in
    tests/test_oracle_ml.py."""
    a = [np.float64(v) for v in a]
    n = len(a)
    if n < 8:
        res = np.float64(-0.0)
        for v in a:
            res = res + v
    else:
        assert n <= 128
        r = a[:8]
        i = 8
        while i < n - (n % 8):
            r = [r[k] + a[i + k] for k in range(8)]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[i]
            i += 1
    return np.float64(0.0) + res
