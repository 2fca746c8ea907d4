"""CPU tests of the ADMM oracle (oracle/admm_oracle.c): its parity-polytope projection against the reference's own
projection.cpp (compiled as oracle/_ref/libppolytope.so) and the whole decoder against vectors captured from the
reference's ADMM class (src/admm.py:42-69; oracle/make_goldens_admm.py)."""
import json
import os

import numpy as np
import pytest

import admm_oracle as A
from helpers import GOLDEN, golden_edges


def admm_cases():
    with open(os.path.join(GOLDEN, "admm_cases.json")) as fp:
        return json.load(fp)


def admm_arrays(case):
    z = np.load(os.path.join(GOLDEN, "admm_vectors.npz"))
    return {k[len(case["tag"]) + 1:]: z[k] for k in z.files if k.startswith(case["tag"] + "_")}


def graph_of(code_name):
    from ldpc_decoders_amd import codes

    try:
        return golden_edges(code_name)
    except KeyError:
        c = codes.get_code(code_name)

        class G:
            m, n, chk, var = c.m, c.n, c.edge_chk, c.edge_var
        return G


@pytest.mark.skipif(A.ref_lib() is None, reason="oracle/_ref/libppolytope.so not built (make -C oracle ref needs /root/reference)")
def test_projection_bit_exact_vs_reference_library():
    rng = np.random.default_rng(0)
    for L in range(1, 17):
        for t in range(1500):
            kind = t % 5
            if kind == 0:
                v = rng.normal(0.5, 0.6, L)
            elif kind == 1:
                v = rng.uniform(-0.2, 1.2, L)
            elif kind == 2:
                v = np.round(rng.uniform(-0.5, 1.5, L) * 4) / 4  # many ties
            elif kind == 3:
                v = rng.normal(0.5, 2.0, L)
            else:
                v = rng.choice([0.0, 1.0, 0.5, 0.25, 0.75, 1.5, -0.5], L)
            assert np.array_equal(A.pp_project(v), A.pp_project_ref(v)), (L, v)


def test_projection_lands_in_the_polytope():
    # size-independent properties: inside the unit cube, every odd-set facet inequality holds, vertices are fixed points
    rng = np.random.default_rng(1)
    for L in (2, 3, 6, 7, 8):
        for _ in range(300):
            p = A.pp_project(rng.normal(0.5, 1.0, L))
            assert (p >= 0).all() and (p <= 1).all()
            s = np.sort(p)[::-1]
            for k in range(1, L + 1, 2):  # most violated facet for each odd set size: the k largest entries
                assert s[:k].sum() - s[k:].sum() <= k - 1 + 1e-9
        for w in range(0, L + 1, 2):
            v = np.zeros(L)
            v[rng.permutation(L)[:w]] = 1
            assert np.array_equal(A.pp_project(v), v)


def test_numpy_sum_order():
    rng = np.random.default_rng(2)
    for n in (1, 7, 8, 9, 127, 128, 129, 1000, 3600, 7920, 29982):
        x = rng.standard_normal(n) ** 2 * 10.0 ** rng.uniform(-6, 2, n)
        assert A.np_sum(x) == x.sum()


@pytest.mark.parametrize("case", admm_cases(), ids=lambda c: "%s-%s-%s" % (c["channel"], c["code"], c["param"]))
def test_admm_oracle_reproduces_reference(case):
    a = admm_arrays(case)
    g = graph_of(case["code"])
    x, iters, conv = A.admm_decode(g, a["gamma"], case["mu"], case["eps"], case["max_iter"])
    assert np.array_equal(iters, a["iters"])
    est = A.pseudo_to_cw(x, case["allow_pseudo"]).astype(np.float64)
    assert np.array_equal(est, a["xhat"], equal_nan=True)  # a degree-0 variable is 0/0 upstream, too
    assert ((iters < case["max_iter"]) == (conv == 1)).all() or case["max_iter"] <= 0
