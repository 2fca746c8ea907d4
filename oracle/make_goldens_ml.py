#!/usr/bin/env python3
"""Golden vectors of the reference's ML decoders (run in the build container only; see oracle/make_goldens.py).

Writes
  tests/golden/ml_vectors.npz    per case: the received words of a seeded (send, decode) sequence, the log-likelihood
                                 matrix the reference computed for each (captured at math_utils.arg_max_rand), the
                                 index it picked and the word it returned
  tests/golden/ml_cases.json     the case list (channel, code, param, codeword, seed, frames)
  tests/golden/ml_kat.json       the reference's six known-answer tests run with its ML classes
  tests/golden/main_counters_ml.json   tot/wec/bec of reference main.py runs with decoder ML under fixed seeds
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

CASES = [  # channel, code, param, codeword (-1 = random words), seed, frames
    ("biawgn", "7_4_hamming", 2.0, 0, 201, 200), ("biawgn", "7_4_hamming", -1.0, 1, 202, 200),
    ("biawgn", "12_3_4_ldpc", 1.0, 0, 203, 200), ("biawgn", "4_2_test", 0.0, -1, 204, 200),
    ("biawgn", "6_2_3_ldpc", 3.0, -1, 205, 100),
    ("bsc", "7_4_hamming", 0.1, 0, 211, 300), ("bsc", "12_3_4_ldpc", 0.15, -1, 212, 300), ("bsc", "4_2_test", 0.3, 1, 213, 100),
    ("bsc", "7_4_hamming", 0.5, 0, 214, 100), ("bsc", "6_2_3_ldpc", 0.7, 0, 215, 100),
    ("bec", "7_4_hamming", 0.3, 0, 221, 300), ("bec", "12_3_4_ldpc", 0.5, -1, 222, 300), ("bec", "4_2_test", 0.4, 1, 223, 100),
    ("bec", "6_2_3_ldpc", 0.6, -1, 224, 100),
]


def gen_vectors(R):
    arrays, meta = {}, []
    for i, (ch, code, param, cw, seed, frames) in enumerate(CASES):
        mod = getattr(R, ch)
        cobj = R.codes.get_code(code)
        chan, dec = mod.Channel(param), mod.ML(param, cobj)
        captured = []
        orig = R.math_utils.arg_max_rand

        def spy(values):
            ind = orig(values)
            captured.append((np.array(values, dtype=np.float64), int(ind)))
            return ind

        R.math_utils.arg_max_rand = spy
        try:
            np.random.seed(seed)
            Y, X, XH = [], [], []
            with np.errstate(all="ignore"):
                for _ in range(frames):
                    x = cobj.cb[np.random.choice(cobj.cb.shape[0], 1)[0]] if cw == -1 else cobj.parity_mtx[0] * 0 + cw
                    y = chan.send(x)
                    xh = dec.decode(y)
                    Y.append(np.array(y)), X.append(np.array(x)), XH.append(np.array(xh))
        finally:
            R.math_utils.arg_max_rand = orig
        tag = "c%02d" % i
        arrays[tag + "_y"] = np.array(Y, dtype=np.float64 if ch == "biawgn" else np.int8)
        arrays[tag + "_x"] = np.array(X, dtype=np.int8)
        arrays[tag + "_xhat"] = np.array(XH, dtype=np.int8)
        arrays[tag + "_log_prob"] = np.array([c[0] for c in captured])
        arrays[tag + "_pick"] = np.array([c[1] for c in captured], dtype=np.int32)
        arrays[tag + "_cb"] = np.array(cobj.cb, dtype=np.int8)
        meta.append(dict(tag=tag, channel=ch, code=code, param=param, codeword=cw, seed=seed, frames=frames))
        ties = np.mean([(c[0] == c[0].max()).sum() for c in captured])
        print("  ml vectors:", meta[-1], "mean tie-set size %.2f" % ties, flush=True)
    np.savez_compressed(os.path.join(GOLD, "ml_vectors.npz"), **arrays)
    with open(os.path.join(GOLD, "ml_cases.json"), "w") as fp:
        json.dump(meta, fp, indent=1)


def gen_kat(R):
    kats = [  # src/biawgn.py:81-92, src/bsc.py Test, src/bec.py Test
        ("biawgn", "4_2_test", 1.0, [1, 1, 0, 1, 1], [1, 1, 1.6, .9, 1]),
        ("biawgn", "7_4_hamming", .1, [1, 0, 0, 1, 1, 0, 0], [1, -1, 1.1, 1, 1, -1, -1]),
        ("bsc", "4_2_test", 1 / 3, [1, 1, 0, 1, 1], [1, 0, 0, 1, 1]),
        ("bsc", "7_4_hamming", .1, [1, 0, 0, 1, 1, 0, 0], [1, 0, 1, 1, 1, 0, 0]),
        ("bec", "4_2_test", 1 / 3, [1, 1, 0, 1, 1], [1, 2, 0, 1, 2]),
        ("bec", "7_4_hamming", .1, [1, 0, 0, 1, 1, 0, 0], [2, 0, 2, 1, 1, 0, 2]),
    ]
    out = []
    for ch, code, param, x, y in kats:
        np.random.seed(0)
        est = getattr(R, ch).ML(param, R.codes.get_code(code), max_iter=100).decode(np.array(y))
        out.append(dict(channel=ch, code=code, param=param, sent=x, received=y, decoder="ML", np_seed=0,
                        reference_estimate=[int(v) for v in est], reference_pass=bool((est == np.array(x)).all())))
        print("  ml kat:", ch, code, out[-1]["reference_pass"], flush=True)
    with open(os.path.join(GOLD, "ml_kat.json"), "w") as fp:
        json.dump(out, fp, indent=1)


def gen_main(R):
    import runpy

    runs = [
        (21, "biawgn 7_4_hamming ML --codeword 0 --min-wec 60 --params 2 4"),
        (22, "biawgn 12_3_4_ldpc ML --codeword -1 --min-wec 40 --params 1.0 3.0"),
        (23, "bsc 7_4_hamming ML --codeword 1 --min-wec 60 --params 0.1 0.05"),
        (24, "bsc 4_2_test ML --codeword -1 --min-wec 40 --params 0.2"),
        (25, "bec 7_4_hamming ML --codeword 0 --min-wec 60 --params 0.3 0.2"),
        (26, "bec 12_3_4_ldpc ML --codeword -1 --min-wec 40 --params 0.5"),
    ]
    out = []
    tmp = "/tmp/ldpc_goldens_main_ml"
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
    with open(os.path.join(GOLD, "main_counters_ml.json"), "w") as fp:
        json.dump(out, fp, indent=1)


if __name__ == "__main__":
    R = ref_import.load()
    gen_vectors(R)
    gen_kat(R)
    gen_main(R)
