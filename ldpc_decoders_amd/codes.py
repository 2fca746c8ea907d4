"""Code store: parity-check matrices as sparse row-major edge lists.

Host-side mirror of the reference's ``src/codes.py`` for the BP hot path: same names
(``Code``, ``get_code``, ``get_code_names``, ``load_parity_mtx``, ``save_parity_mtx``, ``rand_reg_ldpc``), same
built-in toy codes (src/codes.py:27-66), same text format and loader semantics (src/codes.py:93-105) -- but H is
never held dense (the reference's dense int64 H is 16.8 GB at n = 64 800): a code is its edge list
``(edge_chk[k], edge_var[k])`` in the order of ``np.where(H)``.  ``parity_mtx`` is still offered (built lazily) so
that code written against the reference keeps working for small codes.
"""
import itertools
import os

import numpy as np

file_codes_dir_string = "FILE_CODES_DIR"


PACKAGE_CODES_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "codes")  # the reference's 27 data/codes files


def _file_codes_dir():
    """src/codes.py:68-70: $FILE_CODES_DIR, else ./data/codes relative to the working directory -- and, where upstream would
    find nothing (it is always run from its own checkout), the copies of its data/codes files that ship inside this package."""
    env = os.environ.get(file_codes_dir_string)
    if env:
        return os.path.abspath(env)
    cwd = os.path.abspath(os.path.join("data", "codes"))
    return cwd if os.path.isdir(cwd) else PACKAGE_CODES_DIR


class Code:
    """A binary linear block code given by its parity checks (and optionally generator rows).

    reference: ``codes.Code`` (src/codes.py:8-24).  Accepts either dense matrices (reference signature
    ``Code(gen_mtx, parity_mtx)``) or, via :meth:`from_edges` / :meth:`from_rows`, sparse descriptions.
    """

    def __init__(self, gen_mtx=None, parity_mtx=None):
        self.gen_mtx = None if gen_mtx is None else np.asarray(gen_mtx, dtype=np.int64)
        self._dense = None
        self._cb = None
        self._handles = {}
        if parity_mtx is not None:
            H = np.asarray(parity_mtx)
            chk, var = np.nonzero(H)
            self._set_edges(H.shape[0], H.shape[1], chk, var)

    def _set_edges(self, m, n, chk, var):
        chk = np.ascontiguousarray(chk, dtype=np.int32)
        var = np.ascontiguousarray(var, dtype=np.int32)
        order = np.lexsort((var, chk))
        chk, var = chk[order], var[order]
        if len(chk) > 1:
            dup = (np.diff(chk) == 0) & (np.diff(var) == 0)
            if dup.any():
                keep = np.r_[True, ~dup]
                chk, var = chk[keep], var[keep]
        self.m, self.n, self.E = int(m), int(n), int(len(chk))
        self.edge_chk, self.edge_var = np.ascontiguousarray(chk), np.ascontiguousarray(var)
        return self

    @classmethod
    def from_edges(cls, m, n, chk, var, gen_mtx=None):
        return cls(gen_mtx)._set_edges(m, n, chk, var)

    @classmethod
    def from_rows(cls, n, rows, gen_rows=None):
        """rows: iterable of iterables with the variable indices (0-based) of each check."""
        chk = [c for c, r in enumerate(rows) for _ in r]
        var = [v for r in rows for v in r]
        gen = None
        if gen_rows is not None:
            gen = np.zeros((len(gen_rows), n), dtype=np.int64)
            for i, r in enumerate(gen_rows):
                gen[i, list(r)] = 1
        return cls(gen)._set_edges(len(rows), n, chk, var)

    # ---- reference-compatible surface
    @property
    def parity_mtx(self):
        if self._dense is None:
            if self.m * self.n > 1 << 28:
                raise MemoryError("dense H of %dx%d refused; use the edge list (edge_chk, edge_var)" % (self.m, self.n))
            H = np.zeros((self.m, self.n), dtype=np.int64)
            H[self.edge_chk, self.edge_var] = 1
            self._dense = H
        return self._dense

    @property
    def cb(self):
        """Code book, all 2^k words (small codes only; reference builds it eagerly, src/codes.py:12-19)."""
        if self._cb is None:
            if self.gen_mtx is None:
                raise AttributeError("code has no generator matrix, hence no code book")
            k = self.gen_mtx.shape[0]
            msgs = np.array(list(itertools.product((0, 1), repeat=k)), dtype=np.int64)
            cb = (msgs @ self.gen_mtx) % 2
            assert self.syndrome(cb).sum() == 0 and cb[0].sum() == 0
            self._cb = cb
        return self._cb

    def syndrome(self, words):
        """(H @ words^T) mod 2 without a dense H; words [..., n] -> [..., m]."""
        w = np.asarray(words)
        s = np.zeros(w.shape[:-1] + (self.m,), dtype=np.int64)
        np.add.at(s, (Ellipsis, self.edge_chk), w[..., self.edge_var].astype(np.int64))
        return s % 2

    def get_n(self):
        return self.n

    def get_k(self):
        return self.n - self.m

    def row_degrees(self):
        return np.bincount(self.edge_chk, minlength=self.m)

    def col_degrees(self):
        return np.bincount(self.edge_var, minlength=self.n)


# Built-in toy codes (the matrices of src/codes.py:27-66, written as row supports).
_BUILTIN = {
    "4_2_test": (5, [(0, 1), (1, 2, 3), (3, 4)], [(0, 1, 2), (2, 3, 4)]),
    "6_2_3_ldpc": (6, [(0, 1, 2), (3, 4, 5), (2, 3, 5), (0, 1, 4)], [(3, 5), (0, 2, 3, 4), (0, 1)]),
    "7_4_hamming": (7, [(3, 4, 5, 6), (1, 2, 5, 6), (0, 2, 4, 6)], [(0, 1, 2), (0, 3, 4), (1, 3, 5), (0, 1, 3, 6)]),
    "12_3_4_ldpc": (12,
                    [(2, 5, 6, 7), (0, 1, 4, 11), (3, 8, 9, 10), (1, 5, 6, 9), (0, 2, 7, 10), (3, 4, 8, 11), (0, 3, 4, 6),
                     (5, 7, 10, 11), (1, 2, 8, 9)],
                    [(4, 5, 6, 11), (3, 6, 7, 8, 9, 10), (2, 5, 9, 10), (1, 5, 7, 8, 10, 11), (0, 6, 7, 8, 9, 11)]),
}


def get_file_code_map():
    d = _file_codes_dir()
    files = next(os.walk(d), ((), (), ()))[2]
    return {os.path.splitext(f)[0]: os.path.join(d, f) for f in files}


def get_code_names():
    return list(_BUILTIN.keys()) + list(get_file_code_map().keys())


def get_code(name):
    """reference: ``codes.get_code`` (src/codes.py:84-90): files shadow built-ins."""
    files = get_file_code_map()
    if name in files:
        return load_parity_mtx(files[name])
    n, rows, gen_rows = _BUILTIN[name]
    return Code.from_rows(n, rows, gen_rows)


def parse_parity_text(text):
    """Text -> Code with the loader semantics of src/codes.py:93-105.

    One check per non-blank line, whitespace-separated variable numbers; the index base is the global minimum,
    which must be 0 or 1; ``n = max + (0 if base == 1 else 1)``.  Quirk kept for bit-compatibility: every entry is
    written to column ``var - 1`` whatever the base, so in a 0-based file variable 0 lands in the LAST column.
    """
    rows = [[int(t) for t in line.split()] for line in text.splitlines() if line.split()]
    if not rows:
        raise Exception("empty parity file")
    lo = min(min(r) for r in rows)
    hi = max(max(r) for r in rows)
    if lo not in (0, 1):
        raise Exception("Minimum index is not 0 or 1.")
    n = hi + (0 if lo == 1 else 1)
    chk = np.repeat(np.arange(len(rows)), [len(r) for r in rows])
    var = (np.concatenate([np.asarray(r, dtype=np.int64) for r in rows]) - 1) % n
    return Code.from_edges(len(rows), n, chk, var)


def load_parity_mtx(file_path):
    with open(file_path, "r") as fp:
        return parse_parity_text(fp.read())


def save_parity_mtx(code, code_name, directory=None):
    """reference: ``codes.save_parity_mtx`` (src/codes.py:131-136): 1-based indices, one check per line."""
    if not isinstance(code, Code):
        code = Code(None, code)
    directory = directory or _file_codes_dir()
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "%s.txt" % code_name)
    starts = np.r_[0, np.cumsum(code.row_degrees())]
    with open(path, "w") as fp:
        for c in range(code.m):
            fp.write(" ".join(str(v + 1) for v in code.edge_var[starts[c]:starts[c + 1]]) + "\n")
    return path


def rand_reg_ldpc(n, l, r, rng=None):
    """Random (l, r)-regular code, same ensemble as ``codes.rand_reg_ldpc`` (src/codes.py:108-120): every check takes
    the r currently least-loaded variables, ties broken uniformly at random.  Sparse (no dense H).

    When r divides n the greedy fill is exactly "l independent uniform permutations of the variables, dealt r at a
    time" (all variables of the current minimum degree are equally likely, and a level is used up exactly at a check
    boundary), which is what is done here in O(E); otherwise the level boundaries are handled check by check.
    """
    rng = rng or np.random
    m = int(n * l / r)
    if n % r == 0 and m * r == n * l:
        var = np.concatenate([rng.permutation(n) for _ in range(l)])
        chk = np.repeat(np.arange(m), r)
    else:
        deg = np.zeros(n, dtype=np.int64)
        chk, var = [], []
        for i in range(m):
            key = deg + rng.random_sample(n)  # integer part orders by load, fractional part breaks ties at random
            pick = np.argpartition(key, r - 1)[:r]
            deg[pick] += 1
            chk += [i] * r
            var += [int(v) for v in pick]
    code = Code.from_edges(m, n, chk, var)
    assert (code.col_degrees() == l).all() and (code.row_degrees() == r).all()
    return code


def irregular_degree_counts(n, lambda_edge, dc):
    """Variables per degree of the irregular ensemble -> ({degree: count}, extra degrees).

    ``floor(L_i n)`` variables of degree i with L_i ~ lambda_i / i (node perspective, ``get_node_dist`` / ``gen_L_R`` of
    src/ldpc.py:138-143,158-159 -- pinned to the reference's counts by tests/golden/irregular_ensembles.json), then the few
    variables still missing.  Upstream hard-codes those (``extra``, src/ldpc.py:154,167 -- which is why it asserts out for
    n != 1200); here they are solved for: the smallest combination of ensemble degrees that makes the socket count a multiple of dc."""
    degs = sorted(lambda_edge)
    node = np.array([lambda_edge[d] / d for d in degs], dtype=np.float64)
    node /= node.sum()
    counts = {d: int(f * n) for d, f in zip(degs, node)}
    left = n - sum(counts.values())
    sockets = sum(d * c for d, c in counts.items())
    for combo in itertools.combinations_with_replacement(degs, left):
        if (sockets + sum(combo)) % dc == 0:
            for d in combo:
                counts[d] += 1
            return counts, list(combo)
    raise ValueError("cannot complete the degree sequence for n=%d, dc=%d" % (n, dc))


def rand_irregular_ldpc(n, lambda_edge, dc, rng=None):
    """Random irregular code from an edge-perspective variable degree distribution and check degree ``dc``.

    Same ensemble as ``ldpc.gen_rand_irg_ldpc`` (src/ldpc.py:149-192): node fractions L_i ~ lambda_i / i, floor(L_i n)
    variables of degree i, variable sockets matched to check sockets by a uniform permutation, edges of even
    multiplicity cancelled (src/ldpc.py:189).  The reference hard-codes the few "extra" variables that make the counts
    integral (src/ldpc.py:154,167) and therefore asserts out for n != 1200; here they are solved for: the leftover
    variables take the degrees that make the socket count a multiple of dc.  ``lambda_edge``: {degree: edge fraction}.
    Sparse, O(E).  Degree-0/odd-multiplicity effects mean a few checks end up with degree < dc, as upstream.
    """
    rng = rng or np.random
    counts, _ = irregular_degree_counts(n, lambda_edge, dc)
    degs = sorted(counts)
    var_sockets = np.concatenate([np.repeat(np.arange(start, start + counts[d]), d)
                                  for d, start in zip(degs, np.cumsum([0] + [counts[d] for d in degs[:-1]]))])
    m = len(var_sockets) // dc
    chk_sockets = np.tile(np.arange(m), dc)  # check sockets in the order 1..m repeated dc times, as upstream
    var_sockets = var_sockets[rng.permutation(len(var_sockets))]
    key = chk_sockets.astype(np.int64) * n + var_sockets
    uniq, mult = np.unique(key, return_counts=True)
    keep = uniq[mult % 2 == 1]
    return Code.from_edges(m, n, keep // n, keep % n)


# lambda(x) of the reference's LP design for rho(x) = x^5, rate 1/2 (from ldpc.solve_dist src/ldpc.py:83-94)
LAMBDA_RHO_X5_HALF_RATE = {2: 0.4126, 3: 0.1763, 4: 0.1189, 7: 0.1136, 8: 0.1786}
# the same design for the other check degrees `python src/ldpc.py irg --rho R --rate .5` accepts (probed from ldpc.solve_dist; the
# full values are in tests/golden/irregular_ensembles.json): {rho exponent: {variable degree: edge fraction}}, check degree = rho + 1
LAMBDA_HALF_RATE = {
    4: {2: 0.55215011, 3: 0.14354961, 4: 0.30430028},
    5: LAMBDA_RHO_X5_HALF_RATE,
    6: {2: 0.33871274, 3: 0.14130372, 4: 0.10088889, 6: 0.09617581, 7: 0.09710031, 15: 0.00480352, 16: 0.22101500},
}


def gen_rand_ldpc(args):
    """CLI of the reference's code generators: ``python src/codes.py <count> <n> <l> <r>`` (src/codes.py:139-174) writes
    ``<n>_<l>_<r>_rand_ldpc_<i>.txt``; with ``--irregular [--rho R]`` the rate-1/2 rho = x^R ensemble of ``python src/ldpc.py irg
    --rho R --rate .5`` (src/ldpc.py:149-192) as ``<n>_rho_x<R>_rand_ldpc_<i>.txt``.  Files land in $FILE_CODES_DIR (default data/codes)."""
    out = []
    for i in range(args.count):
        if args.irregular:
            code = rand_irregular_ldpc(args.n, LAMBDA_HALF_RATE[args.rho], args.rho + 1)
            name = "%d_rho_x%d_rand_ldpc_%d" % (args.n, args.rho, i + 1)
        else:
            code = rand_reg_ldpc(args.n, args.l, args.r)
            name = "%d_%d_%d_rand_ldpc_%d" % (args.n, args.l, args.r, i + 1)
        path = save_parity_mtx(code, name, args.dir)
        chk = load_parity_mtx(path)  # verify_rand_reg_ldpc (src/codes.py:148-152): reload and report the degrees
        print(name, (chk.m, chk.n), sorted(set(chk.col_degrees().tolist())), sorted(set(chk.row_degrees().tolist())))
        out.append(path)
    return out


def setup_parser():
    import argparse

    p = argparse.ArgumentParser()
    p.add_argument("count", help="number of random codes to generate", type=int)
    p.add_argument("n", help="code length", type=int)
    p.add_argument("l", help="variable degree of the regular ensemble", type=int, nargs="?", default=3)
    p.add_argument("r", help="check degree of the regular ensemble", type=int, nargs="?", default=6)
    p.add_argument("--irregular", action="store_true", help="rate-1/2 irregular ensemble (lambda of the reference's LP design for rho = x^RHO)")
    p.add_argument("--rho", type=int, default=5, choices=sorted(LAMBDA_HALF_RATE), help="exponent of rho(x) = x^RHO, i.e. check degree RHO + 1 (with --irregular)")
    p.add_argument("--dir", default=None, help="output directory (default: $FILE_CODES_DIR or data/codes)")
    return p


if __name__ == "__main__":
    gen_rand_ldpc(setup_parser().parse_args())
