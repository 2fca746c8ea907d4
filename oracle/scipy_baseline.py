"""Per-frame scipy.sparse min-sum / sum-product decoder: the CPU baseline of SURVEY.md section 8(d).

TEST / MEASUREMENT INFRASTRUCTURE -- never imported by the product package (only bench.py's `cpu_baseline` leg, oracle/make_timing.py
and tests/ use it).  It is the build's own statement of the reference's algorithm (thadikari/ldpc_decoders src/bpa.py:17-102,
src/math_utils.py:7-60) in the reference's own vocabulary of operations -- one frame at a time, scipy.sparse containers over the
row-major edge list, `sum(axis=...)` reductions, numpy transcendental functions -- so that its frames/s per core stand for "the
reference's class of CPU implementation" on hosts where the reference itself cannot run (the GPU box has no /root/reference).
How close it is to the real thing is MEASURED where both can run (oracle/make_timing.py -> tests/golden/reference_timing.json:
`calibration` = reference frames/s / this file's frames/s on identical frames).  Differences from the reference, on purpose: H is
never dense (the syndrome test is a sparse product), and the two row minima come from `np.minimum.reduceat` over the CSR data instead
of CSR fancy indexing -- the reference's own comment calls its min-sum update "way slower than SPA" (src/bpa.py:87).

Results equal oracle/bp_oracle.py bit for bit (tests/test_oracle_golden.py::test_scipy_baseline_equals_the_oracle).
"""
import numpy as np
import scipy.sparse as sp


class ScipyBP:
    def __init__(self, m, n, chk, var, alg="MSA", max_iter=50):
        self.m, self.n, self.alg, self.max_iter = int(m), int(n), alg, int(max_iter)
        self.chk, self.var = np.asarray(chk, dtype=np.int64), np.asarray(var, dtype=np.int64)  # row-major edge order (np.where(H))
        self.E = len(self.chk)
        self.H = sp.csr_matrix((np.ones(self.E, dtype=np.int64), (self.chk, self.var)), shape=(m, n))
        self.row_ptr = self.H.indptr.astype(np.int64)
        self.row_start = self.row_ptr[:-1][np.diff(self.row_ptr) > 0]
        self.row_len = np.diff(self.row_ptr)
        self.nonempty = self.row_len > 0
        self.iterations = 0

    def _coo(self, data):
        return sp.coo_matrix((data, (self.chk, self.var)), shape=(self.m, self.n))

    def _col_sum(self, data):  # src/math_utils.py:7 through src/bpa.py:15: COO column sum
        return np.asarray(self._coo(data).sum(axis=0)).ravel()

    def _row_sum(self, data):
        return np.asarray(self._coo(data).sum(axis=1)).ravel()

    def _row_parity_sign(self, v):  # src/math_utils.py:38-43
        return (self._row_sum((v < 0).astype(np.int64)) % 2) * -2 + 1

    def _check_msa(self, v2c):  # src/bpa.py:86-102
        sign = self._row_parity_sign(v2c)[self.chk] / ((v2c >= 0).astype(np.int64) * 2 - 1)
        a = np.abs(v2c)
        min1 = np.full(self.m, np.inf)
        min1[self.nonempty] = np.minimum.reduceat(a, self.row_start)
        hit = np.flatnonzero(a == min1[self.chk])                       # every edge at its row minimum ...
        first = hit[np.searchsorted(hit, self.row_start)]                # ... the FIRST one per row (ties: lowest edge)
        b = a.copy()
        b[first] = np.inf
        min2 = np.full(self.m, np.inf)
        min2[self.nonempty] = np.minimum.reduceat(b, self.row_start)
        mag = min1[self.chk]
        mag[first] = min2[self.nonempty]
        return sign * mag

    def _check_spa(self, v2c):  # src/bpa.py:71-75, src/math_utils.py:47-60
        t = np.tanh(v2c / 2.0)
        prod = self._row_parity_sign(t) * np.exp(self._row_sum(np.log(np.abs(t))))
        q = prod[self.chk] / t
        out = np.empty_like(q)
        one = np.abs(q) == 1
        out[one] = np.inf * q[one]
        out[~one] = np.arctanh(q[~one])
        return 2 * out

    def decode(self, y, priors):
        """One frame, src/bpa.py:17-63: returns x_hat (y itself when the iteration-0 test passes)."""
        v2c = priors[self.var]
        x_hat, it = y, 0
        with np.errstate(all="ignore"):
            while True:
                if 0 < self.max_iter <= it:
                    break
                if ((self.H @ x_hat) % 2 == 0).all():
                    break
                c2v = self._check_msa(v2c) if self.alg == "MSA" else self._check_spa(v2c)
                marginal = priors + self._col_sum(c2v)
                v2c = marginal[self.var] - c2v
                marginal[np.isnan(marginal)] = 0.0
                x_hat = (marginal < 0).astype(np.int64)
                it += 1
        self.iterations = it
        return x_hat
