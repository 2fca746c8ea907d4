"""The shipped LDS layout plans against the matrices they were made for: every code file of the reference (ldpc_decoders_amd/data/codes, the
reference's data/codes) constructs its decoders from a stored plan -- no annealing at construction, few bank-conflict cycles -- and
decodes bit-identically to the C oracle in the reference's arithmetic."""
import glob
import os

import numpy as np
import pytest

import bp_oracle as O
import c_oracle as C
from helpers import CODES_DIR, GOLDEN

pytestmark = pytest.mark.gpu
NAMES = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(CODES_DIR, "*.txt")))


def test_all_reference_code_files_are_fixtures():
    assert len(NAMES) == 27 and sum(n.startswith("1200_3_6_rand_ldpc_") for n in NAMES) == 10 and sum(n.startswith("1200_rho_x5_") for n in NAMES) == 10


@pytest.mark.parametrize("name", NAMES)
def test_stored_plan_and_parity(name, monkeypatch, tmp_path):
    from ldpc_decoders_amd import bpa, codes

    monkeypatch.setenv("LDPC_FUSED_PLAN_SAVE", "none")            # nothing may be annealed-and-kept here ...
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "empty"))  # ... nor found in a user cache: only the shipped plans count
    monkeypatch.setenv("LDPC_FUSED_PLAN_MOVES", "1000")            # a plan that is NOT in the store would show up with hundreds of conflict cycles
    code = codes.load_parity_mtx(os.path.join(CODES_DIR, name + ".txt"))

    class G:
        m, n, chk, var = code.m, code.n, code.edge_chk, code.edge_var

    rng = np.random.RandomState(len(name))
    snr = 2.0
    y = -1 + rng.normal(0, np.sqrt(O.biawgn_noise_var(snr)), (48, code.n))
    pri = O.biawgn_priors(y, snr)
    # n = 512: one wave per frame, 96 gather cycles per sweep.  The irregular codes run on the shapes with pair rounds: 256 gather cycles per
    # sweep instead of 280, for which their tighter slot classes cost ~20 more conflict cycles than the two-width shapes' plans (14-30)
    limit = 60 if name.startswith("512_") or "rho" in name else 40
    for prec, dt in (("f64", np.float64), ("f32", np.float32)):
        dec = bpa.MSA(code, max_iter=40, precision=prec, backend="fused")
        fi = dec.handle.fused_info()
        assert fi["waves_per_frame"] > 0
        assert fi["conflict_cycles_planned"] <= limit < fi["conflict_cycles_identity"], (name, prec, fi)
        xhat, iters = dec.decode_batch(None, pri.astype(dt))
        xo, io = C.bp_decode(G, "MSA", None, pri.astype(dt), 40, dtype=dt)
        assert (xhat == xo).all() and (iters == io).all()
    # fp64 sum-product: the plan keeps every check's edge order (its row sum is order dependent) and is in the store as well
    dec = bpa.SPA(code, max_iter=10, precision="f64", backend="fused")
    fi = dec.handle.fused_info()
    assert fi["conflict_cycles_planned"] < 0.4 * fi["conflict_cycles_identity"], (name, fi)
    ref = bpa.SPA(code, max_iter=10, precision="f64", backend="stream")
    a, b = dec.decode_batch(None, pri[:16]), ref.decode_batch(None, pri[:16])
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()  # same device function, same edge order: bit-identical
