#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference; see oracle/ref_import.py for
the two in-memory shims).  The outputs are data: seeds, received words, expected
hard decisions, iteration counts, marginals and Monte-Carlo counters.  No reference
source travels.  Usage:   python oracle/make_goldens.py [--quick] [--out DIR] [--only SUBSTRING ...]
(oracle/regen_all.sh runs this and its sibling generators; tests/test_goldens_regenerate_cpu.py re-runs a few
cases into a scratch directory and compares them with the committed files.)

Fixtures written (this script is the ONLY writer of each)
  tests/golden/codes_edges.npz       edge lists as loaded by the reference loader (src/codes.py:93-105)
  tests/golden/kat.json              the six known-answer tests + reference outputs (src/{biawgn,bsc,bec}.py Test.test_all)
  tests/golden/decode_<tag>.npz      per case: seed, received words, x_hat (packed), iterations, marginal traces
  tests/golden/main_counters.json    tot/wec/bec of reference main.py runs under fixed seeds (src/main.py:22-50)
(the H data files ship in ldpc_decoders_amd/data/codes: oracle/make_goldens_codes.py; the timing calibration
tests/golden/reference_timing.json: oracle/make_timing.py)
"""
import argparse
import io
import json
import os
import shutil
import sys
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

FILE_CODES = ["1200_3_6_rand_ldpc_1", "1200_rho_x5_rand_ldpc_5", "512_3_6_rand_ldpc_2", "margulis", "1200_3_6_ldpc"]
BUILTIN_CODES = ["4_2_test", "6_2_3_ldpc", "7_4_hamming", "12_3_4_ldpc"]


def iter_counter(dec):
    """Count check-node sweeps (calls of BPA.decode_, src/bpa.py:32) and record sum_cols outputs (src/bpa.py:15,35)."""
    inner = getattr(dec, "dec", dec)
    state = {"n": 0, "sums": []}
    if hasattr(inner, "decode_"):
        orig = inner.decode_

        def wrapped(*a, **k):
            state["n"] += 1
            return orig(*a, **k)

        inner.decode_ = wrapped
        orig_sum = inner.sum_cols

        def rec(d):
            out = orig_sum(d)
            state["sums"].append(np.array(out, dtype=np.float64))
            return out

        inner.sum_cols = rec
    return state


def gen_codes(R):
    out = {}
    for name in FILE_CODES + BUILTIN_CODES:
        H = R.codes.get_code(name).parity_mtx
        chk, var = np.where(H)
        out[name + "__shape"] = np.array(H.shape, dtype=np.int32)
        out[name + "__chk"] = chk.astype(np.int32)
        out[name + "__var"] = var.astype(np.int32)
    np.savez_compressed(os.path.join(GOLD, "codes_edges.npz"), **out)


def gen_kat(R):
    kats = [
        ("biawgn", "4_2_test", 1.0, [1, 1, 0, 1, 1], [1, 1, 1.6, .9, 1]),
        ("biawgn", "7_4_hamming", .1, [1, 0, 0, 1, 1, 0, 0], [1, -1, 1.1, 1, 1, -1, -1]),
        ("bsc", "4_2_test", 1 / 3, [1, 1, 0, 1, 1], [1, 0, 0, 1, 1]),
        ("bsc", "7_4_hamming", .1, [1, 0, 0, 1, 1, 0, 0], [1, 0, 1, 1, 1, 0, 0]),
        ("bec", "4_2_test", 1 / 3, [1, 1, 0, 1, 1], [1, 2, 0, 1, 2]),
        ("bec", "7_4_hamming", .1, [1, 0, 0, 1, 1, 0, 0], [2, 0, 2, 1, 1, 0, 2]),
    ]
    out = []
    for ch, code, param, x, y in kats:
        for alg in ("SPA", "MSA"):
            dec = getattr(getattr(R, ch), alg)(param, R.codes.get_code(code), max_iter=100)
            with np.errstate(all="ignore"):
                est = dec.decode(np.array(y))
            out.append(dict(channel=ch, code=code, param=param, sent=x, received=y, decoder=alg, max_iter=100,
                            reference_estimate=[float(v) for v in est], reference_pass=bool((est == np.array(x)).all())))
    with open(os.path.join(GOLD, "kat.json"), "w") as fp:
        json.dump(out, fp, indent=1)


def gen_decode_case(R, tag, ch, alg, code, param, cw, nframes, max_iter, seed, ntrace=2, trace_iters=3):
    mod = getattr(R, ch)
    cobj = R.codes.get_code(code)
    H = cobj.parity_mtx
    x = H[0] * 0 + cw
    chan = mod.Channel(param)
    dec = getattr(mod, alg)(param, cobj, max_iter=max_iter)
    st = iter_counter(dec)
    np.random.seed(seed)
    Y, X, IT, TR = [], [], [], []
    for i in range(nframes):
        y = chan.send(x)
        st["n"], st["sums"] = 0, []
        with np.errstate(all="ignore"):
            xh = dec.decode(y)
        Y.append(np.array(y))
        X.append(np.array(xh, dtype=np.float64))
        IT.append(st["n"])
        if i < ntrace and ch != "bec":
            tr = np.full((trace_iters, H.shape[1]), np.nan)
            for j, s in enumerate(st["sums"][:trace_iters]):
                tr[j] = s  # sum of check->variable messages per variable (marginal - prior)
            TR.append(tr)
    Y = np.array(Y)
    X = np.array(X)
    # x_hat is 0/1(/2) except for frames returned at iteration 0 over BI-AWGN (raw y): keep those rows in full
    raw_rows = np.flatnonzero(~np.isin(X, (0.0, 1.0, 2.0)).all(axis=1))
    np.savez_compressed(
        os.path.join(GOLD, "decode_%s.npz" % tag),
        channel=ch, decoder=alg, code=code, param=param, codeword=cw, max_iter=max_iter, seed=seed, nframes=nframes,
        y=Y.astype(np.float64 if ch == "biawgn" else np.uint8),
        xhat=np.where(np.isin(X, (0.0, 1.0, 2.0)), X, 255).astype(np.uint8),
        raw_rows=raw_rows, iters=np.array(IT, dtype=np.int32),
        sumcols_trace=np.array(TR) if TR else np.zeros((0, trace_iters, H.shape[1])),
    )
    nerr = int(((X != x).sum(axis=1) > 0).sum())
    print("  %-44s frames=%d word-errors=%d mean-iters=%.2f" % (tag, nframes, nerr, float(np.mean(IT))), flush=True)


def gen_main_counters(R, quick):
    """Run the reference CLI driver in-process under a fixed global seed (src/main.py:67-69 leaves it unseeded)."""
    import runpy

    runs = [
        (1234, "biawgn 1200_3_6_rand_ldpc_1 MSA --codeword 0 --min-wec 5 --max-iter 50 --params 2.0"),
        (1234, "biawgn 1200_3_6_rand_ldpc_1 SPA --codeword 0 --min-wec 5 --max-iter 50 --params 1.5"),
        (1234, "bsc 1200_3_6_rand_ldpc_1 SPA --codeword 0 --min-wec 5 --max-iter 50 --params 0.07"),
        (1234, "bsc 1200_3_6_rand_ldpc_1 MSA --codeword 1 --min-wec 5 --max-iter 50 --params 0.03"),
        (1234, "bec 1200_3_6_rand_ldpc_1 SPA --codeword 0 --min-wec 5 --max-iter 50 --params 0.4"),
        (1234, "biawgn 1200_rho_x5_rand_ldpc_5 MSA --codeword 0 --min-wec 5 --max-iter 50 --params 1.5"),
        (0, "biawgn 7_4_hamming SPA --codeword 1 --min-wec 50 --max-iter 10 --params 2 4"),
        (5, "bsc 7_4_hamming MSA --codeword 0 --min-wec 30 --max-iter 10 --params 0.1 0.05"),
        (5, "bec 7_4_hamming MSA --codeword 1 --min-wec 30 --max-iter 10 --params 0.3 0.2"),
        (11, "biawgn 12_3_4_ldpc MSA --codeword 0 --min-wec 40 --max-iter 20 --params 1.0 3.0"),
        # several --params per run at n = 1200: the reference consumes exactly n draws of the global np.random stream per frame
        # (src/main.py:37-40), so every value after the first starts where the previous one stopped -- pins that hand-over for the
        # chunked --exact mode of ldpc_decoders_amd.montecarlo
        (77, "biawgn 1200_3_6_rand_ldpc_1 MSA --codeword 0 --min-wec 5 --max-iter 50 --params 1.5 2.0 2.25"),
        (78, "bsc 1200_3_6_rand_ldpc_1 MSA --codeword 0 --min-wec 4 --max-iter 20 --params 0.06 0.05 0.045"),
        (79, "bec 1200_3_6_rand_ldpc_1 SPA --codeword 0 --min-wec 4 --max-iter 50 --params 0.42 0.4"),
    ]
    if quick:
        runs = runs[6:10]
    out = []
    tmp = "/tmp/ldpc_goldens_main"
    shutil.rmtree(tmp, ignore_errors=True)
    for seed, line in runs:
        argv = line.split() + ["--data_dir", tmp, "--console"]
        old = sys.argv
        sys.argv = ["main.py"] + argv
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
    with open(os.path.join(GOLD, "main_counters.json"), "w") as fp:
        json.dump(out, fp, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="small subset (for checking the script itself)")
    ap.add_argument("--out", default=None, help="write the fixtures here instead of tests/golden/")
    ap.add_argument("--only", nargs="*", default=None, help="decode cases whose tag contains one of these strings; nothing else is written")
    args = ap.parse_args()
    global GOLD
    if args.out:
        GOLD = os.path.abspath(args.out)
    R = ref_import.load()
    os.makedirs(GOLD, exist_ok=True)
    if args.only is None:
        print("codes + KATs")
        gen_codes(R)
        gen_kat(R)
    print("decode vectors")
    big, small = (8, 40) if args.quick else (40, 300)
    cases = []
    for alg in ("MSA", "SPA"):
        for snr in (1.0, 2.0, 3.0):
            cases.append(("biawgn", alg, "1200_3_6_rand_ldpc_1", snr, 0, big, 50))
        cases.append(("biawgn", alg, "1200_rho_x5_rand_ldpc_5", 2.0, 0, big, 50))
        cases.append(("biawgn", alg, "1200_3_6_rand_ldpc_1", 2.5, 1, big, 10))
        cases.append(("biawgn", alg, "512_3_6_rand_ldpc_2", 2.5, 0, big, 30))
        cases.append(("biawgn", alg, "margulis", 2.0, 0, max(big // 4, 4), 20))
        cases.append(("bsc", alg, "1200_3_6_rand_ldpc_1", 0.04 if alg == "MSA" else 0.07, 0, big, 50))
        cases.append(("bsc", alg, "1200_rho_x5_rand_ldpc_5", 0.03 if alg == "MSA" else 0.06, 0, big, 50))
        cases.append(("bsc", alg, "1200_3_6_rand_ldpc_1", 0.03, 1, big, 20))
        for code in ("4_2_test", "7_4_hamming", "12_3_4_ldpc", "6_2_3_ldpc"):
            cases.append(("biawgn", alg, code, 2.0, 0, small, 10))
            cases.append(("bsc", alg, code, 0.1, 0, small, 10))
    for code, p, cw, nf, mi in [("1200_3_6_rand_ldpc_1", 0.35, 0, big, 50), ("1200_3_6_rand_ldpc_1", 0.42, 0, big, 50),
                                 ("1200_3_6_rand_ldpc_1", 0.40, 1, big, 4), ("1200_rho_x5_rand_ldpc_5", 0.40, 0, big, 50),
                                 ("7_4_hamming", 0.3, 0, small, 10), ("7_4_hamming", 0.3, 1, small, 10),
                                 ("4_2_test", 0.4, 0, small, 10), ("12_3_4_ldpc", 0.35, 0, small, 10)]:
        cases.append(("bec", "SPA", code, p, cw, nf, mi))
    cases.append(("bec", "MSA", "1200_3_6_rand_ldpc_1", 0.38, 0, big, 50))
    for i, (ch, alg, code, param, cw, nf, mi) in enumerate(cases):
        tag = "%s_%s_%s_%s_cw%d_it%d" % (ch, alg, code, str(param).replace(".", "p"), cw, mi)
        if args.only is not None and not any(sub in tag for sub in args.only):
            continue
        gen_decode_case(R, tag, ch, alg, code, param, cw, nf, mi, seed=1000 + i)  # the seed is the case's position in the FULL list
    if args.only is None:
        print("main-loop counters")
        gen_main_counters(R, args.quick)


if __name__ == "__main__":
    main()
