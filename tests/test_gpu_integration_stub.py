"""INTEGRATION.md section B, executed: the `src/bpa_hip.py` a maintainer of the reference would drop next to `src/bpa.py` is cut out of
the document VERBATIM (only the library path placeholder is filled in), exec'd, and driven the way the reference's channel wrappers
drive `bpa.SPA` / `bpa.MSA` (src/biawgn.py:21-42, src/bsc.py:19-39): `Dec(parity_mtx, max_iter=..).decode(y, priors)`.  Checked on the
reference's own known-answer rows (src/biawgn.py:81-92, src/bsc.py:78-89; tests/golden/kat.json) and on the iteration-0 rule of
src/bpa.py:20,29 (a received BSC word that is a codeword comes back as the very same object)."""
import os
import re

import numpy as np
import pytest

import bp_oracle as O
from helpers import golden_edges, kat_cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_module():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# src/bpa_hip\.py.*?)```", text, re.S)
    assert m, "INTEGRATION.md section B: the bpa_hip.py block is gone"
    src = m.group(1)
    assert "/path/to/ldpc_decoders_amd/csrc/libldpc_hip.so" in src
    src = src.replace("/path/to/ldpc_decoders_amd/csrc/libldpc_hip.so", os.path.join(ROOT, "ldpc_decoders_amd", "csrc", "libldpc_hip.so"))
    ns = {"__name__": "bpa_hip"}
    exec(compile(src, "INTEGRATION.md:bpa_hip.py", "exec"), ns)
    return ns


def _dense(name):
    g = golden_edges(name)
    h = np.zeros((g.m, g.n), dtype=int)
    h[g.chk, g.var] = 1
    return h


BP_KATS = [k for k in kat_cases() if k["decoder"] in ("SPA", "MSA") and k["channel"] in ("biawgn", "bsc")]


def test_the_stub_in_the_document_covers_the_bp_known_answers():
    assert len(BP_KATS) >= 4, [(k["channel"], k["decoder"]) for k in kat_cases()]


@pytest.mark.parametrize("kat", BP_KATS, ids=lambda k: "%s-%s-%s" % (k["channel"], k["code"], k["decoder"]))
def test_integration_stub_known_answers(kat):
    stub = _stub_module()
    dec = stub[kat["decoder"]](_dense(kat["code"]), max_iter=kat["max_iter"], mu=3., eps=1e-5)  # kwargs = all CLI args upstream (src/main.py:26)
    y = np.array(kat["received"])
    if kat["channel"] == "biawgn":
        priors = O.biawgn_priors(y.astype(np.float64), kat["param"])  # src/biawgn.py:28
    else:
        priors = O.bsc_priors(y.astype(np.int64), kat["param"])       # src/bsc.py:21,25
    est = dec.decode(y, priors)
    assert (np.asarray(est, dtype=float) == np.array(kat["reference_estimate"])).all()
    assert (np.asarray(est) == np.array(kat["sent"])).all()


def test_integration_stub_iteration_zero_returns_the_received_object():
    stub = _stub_module()
    h = _dense("1200_3_6_rand_ldpc_1")
    dec = stub["MSA"](h, max_iter=50)
    y = np.zeros(h.shape[1], dtype=np.int64)  # the all-zero word passes the syndrome test before any sweep (src/bpa.py:20,29)
    assert dec.decode(y, O.bsc_priors(y, 0.05)) is y
    rng = np.random.RandomState(2)
    y2 = (rng.random_sample(h.shape[1]) < 0.02).astype(np.int64)
    x2 = dec.decode(y2, O.bsc_priors(y2, 0.02))
    assert x2 is not y2 and not x2.any()  # 24 flips at p = 0.02: decoded back to the all-zero word
