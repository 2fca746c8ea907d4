#!/usr/bin/env python3
"""Golden vectors of the reference's ADMM decoder (run in the build container only; see oracle/make_goldens.py).

The reference loads its projection through ctypes from `ppolytope.lib` next to exact.py; here `exact.lib_path` is pointed at
oracle/_ref/libppolytope.so, i.e. the reference's own projection.cpp compiled by oracle/Makefile.

Writes
  tests/golden/admm_vectors.npz        per case: received words, the LLR vectors handed to ADMM_Base.decode, the returned
                                       estimate and the iteration count of every frame
  tests/golden/admm_cases.json         the case list
  tests/golden/main_counters_admm.json tot/wec/bec (+ the decoder's iteration histogram) of reference main.py runs with ADMM
"""
import contextlib
import io
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

CASES = [  # channel, code, param, codeword, seed, frames, mu, eps, max_iter, allow_pseudo
    ("biawgn", "7_4_hamming", 2.0, 0, 301, 150, 3.0, 1e-5, 100, 0),
    ("biawgn", "7_4_hamming", 1.0, 1, 302, 150, 3.0, 1e-5, 100, 1),
    ("biawgn", "12_3_4_ldpc", 1.0, 0, 303, 100, 3.0, 1e-5, 300, 0),
    ("bsc", "7_4_hamming", 0.1, 0, 304, 150, 3.0, 1e-5, 100, 0),
    ("bsc", "12_3_4_ldpc", 0.12, 1, 305, 100, 2.0, 1e-4, 50, 1),
    ("bec", "7_4_hamming", 0.3, 0, 306, 150, 3.0, 1e-5, 100, 1),
    ("bec", "4_2_test", 0.4, 1, 307, 100, 3.0, 1e-5, 60, 0),
    ("biawgn", "1200_3_6_rand_ldpc_1", 2.2, 0, 308, 6, 3.0, 1e-5, 300, 0),
    ("bsc", "1200_3_6_rand_ldpc_1", 0.05, 0, 309, 5, 3.0, 1e-5, 200, 0),
    ("bec", "1200_rho_x5_rand_ldpc_5", 0.35, 0, 310, 5, 3.0, 1e-5, 200, 1),
    ("biawgn", "margulis", 2.0, 0, 311, 2, 3.0, 1e-5, 100, 0),
]


def load_ref():
    R = ref_import.load()
    import importlib

    exact = importlib.import_module("parity_polytope.exact")
    exact.lib_path = os.path.join(HERE, "_ref", "libppolytope.so")
    assert os.path.exists(exact.lib_path), "run `make -C oracle ref` first"
    return R


def gen_vectors(R):
    arrays, meta = {}, []
    for i, (ch, code, param, cw, seed, frames, mu, eps, max_iter, ap) in enumerate(CASES):
        mod = getattr(R, ch)
        cobj = R.codes.get_code(code)
        chan = mod.Channel(param)
        dec = mod.ADMM(param, cobj, mu=mu, eps=eps, max_iter=max_iter, allow_pseudo=ap)
        inner = dec.dec
        gammas = []
        orig = inner.decode

        def spy(y, gamma):
            gammas.append(np.array(gamma, dtype=np.float64))
            return orig(y, gamma)

        inner.decode = spy
        x = cobj.parity_mtx[0] * 0 + cw
        np.random.seed(seed)
        Y, XH, IT = [], [], []
        with np.errstate(all="ignore"):
            for _ in range(frames):
                y = chan.send(x)
                before = inner.iter.copy()
                xh = dec.decode(y)
                IT.append(int(np.flatnonzero(inner.iter - before)[0]))
                Y.append(np.array(y, dtype=np.float64))
                XH.append(np.array(xh, dtype=np.float64))
        tag = "a%02d" % i
        arrays[tag + "_y"] = np.array(Y)
        arrays[tag + "_gamma"] = np.array(gammas)
        arrays[tag + "_xhat"] = np.array(XH)
        arrays[tag + "_iters"] = np.array(IT, dtype=np.int32)
        meta.append(dict(tag=tag, channel=ch, code=code, param=param, codeword=cw, seed=seed, frames=frames, mu=mu, eps=eps,
                         max_iter=max_iter, allow_pseudo=ap))
        print("  admm vectors:", meta[-1], "iters min/mean/max", min(IT), float(np.mean(IT)), max(IT),
              "word errors", int((np.array(XH) != x).any(axis=1).sum()), flush=True)
    np.savez_compressed(os.path.join(GOLD, "admm_vectors.npz"), **arrays)
    with open(os.path.join(GOLD, "admm_cases.json"), "w") as fp:
        json.dump(meta, fp, indent=1)


def gen_main(R):
    import runpy

    runs = [
        (31, "biawgn 7_4_hamming ADMM --codeword 0 --min-wec 40 --max-iter 100 --params 2 4"),
        (32, "bsc 7_4_hamming ADMM --codeword 1 --min-wec 40 --max-iter 100 --params 0.1"),
        (33, "bec 7_4_hamming ADMM --codeword 0 --min-wec 40 --max-iter 100 --allow-pseudo --params 0.3"),
        (34, "biawgn 12_3_4_ldpc ADMM --codeword -1 --min-wec 30 --max-iter 200 --mu 2 --eps 1e-4 --params 1.0"),
    ]
    out = []
    tmp = "/tmp/ldpc_goldens_main_admm"
    shutil.rmtree(tmp, ignore_errors=True)
    for seed, line in runs:
        old = sys.argv
        sys.argv = ["main.py"] + line.split() + ["--data_dir", tmp, "--console"]
        np.random.seed(seed)
        try:
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                runpy.run_path(os.path.join(ref_import.REF_ROOT, "src", "main.py"), run_name="__main__")
        finally:
            sys.argv = old
        files = sorted(os.listdir(tmp))
        newest = max(files, key=lambda f: os.path.getmtime(os.path.join(tmp, f)))
        with open(os.path.join(tmp, newest)) as fp:
            data = json.load(fp)
        out.append(dict(seed=seed, argline=line, file_name=newest, result=data))
        print("  main:", line, "->", {k: data[k] for k in ("tot", "wec", "bec")}, flush=True)
    with open(os.path.join(GOLD, "main_counters_admm.json"), "w") as fp:
        json.dump(out, fp, indent=1)


if __name__ == "__main__":
    R = load_ref()
    gen_vectors(R)
    gen_main(R)
