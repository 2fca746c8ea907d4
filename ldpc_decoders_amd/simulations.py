#!/usr/bin/env python
"""Case tables of the published experiments -- counterpart of the reference's ``simulations.py``.

    python -m ldpc_decoders_amd.simulations <CASE>... [args appended to every line]

prints one ``main.py`` argument line per run (same grammar and the same parameter grids as upstream
``simulations.py:27-85``: cases HMG, MAR, REG_BAD, REG_ENS, IREG_ENS), so ``run_sims.sh`` can feed them to
``python -m ldpc_decoders_amd.main``.  Only the lines of the decoders that are not built here (LP, ADMMA) are skipped -- ML and
ADMM run on the GPU like SPA / MSA; with ``--all-decoders`` the output is identical to upstream's, line for line.
"""
import argparse

BUILT = ("SPA", "MSA", "ML", "ADMM")  # everything upstream's tables name except LP (and ADMMA, which they never name)
ERASURE_GRID = ".5 .475 .45 .425 .4 .375 .35 .34 .33 .325 .32 .31 .3"


def _grid(start, step, count):
    return " ".join("%g" % (start + i * step) for i in range(count))


def default_runs(code, max_iter=10, min_wec=100):
    """The five BP runs made for every LDPC code (upstream exc_def_cases, simulations.py:27-39)."""
    tail = ["--max-iter=%d" % max_iter, "--min-wec=%d" % min_wec]
    table = [
        ("bec", "SPA", 0, ERASURE_GRID),
        ("bsc", "MSA", 1, ".081 .0751 .071 .0651 .061 .0551 .051 .0451 .041 .0351 .031 .0251 .021 .0151 .01"),
        ("biawgn", "MSA", 1, ".5 .75 1. 1.25 1.5 1.75 2. 2.2 2.3 2.4 2.5 2.6 2.7 2.8 2.9 3.0"),
        ("bsc", "SPA", 0, _grid(.1, -.01, 7)),
        ("biawgn", "SPA", 0, ".5 .75 1. 1.25 1.5 1.75 2. 2.25 2.5 2.75 3."),
    ]
    return [[ch, code, dec, "--codeword=%d" % cw] + tail + ["--params " + grid] for ch, dec, cw, grid in table]


def case_HMG():
    p_bec = ".5 .4 .3 .2 .1 .08 .06 .04 .02"
    p_bsc = p_bec + " .25 .15 .01 .008 .006 .004 .002"
    tail = ["--codeword=1", "--min-wec=300"]
    runs = [["bec", "7_4_hamming", d, "--params " + p_bec] + tail for d in ("ML", "LP", "SPA", "ADMM")]
    runs += [["bsc", "7_4_hamming", d, "--params " + p_bsc] + tail for d in ("ML", "LP", "SPA", "MSA", "ADMM")]
    runs += [["biawgn", "7_4_hamming", d, "--params " + _grid(2, .5, 11)] + tail for d in ("ML", "LP", "SPA", "MSA", "ADMM")]
    return runs


def case_MAR():
    tail = ["--codeword=1", "--min-wec=100"]
    runs = [["bec", "margulis", "ADMM", "--params " + ERASURE_GRID] + tail,
            ["bsc", "margulis", "ADMM", "--params .1 .09 .08 .07 .06 .05 .04"] + tail,
            ["biawgn", "margulis", "ADMM", "--params .5 .75 1. 1.25 1.5 1.75 2. 2.25 2.5 2.75 3.0"] + tail]
    return runs + default_runs("margulis")


def case_REG_BAD():
    runs = default_runs("1200_3_6_ldpc")
    for mi in (0, 1, 2, 3, 6, 40, 100):
        runs += default_runs("1200_3_6_ldpc", mi)
    return runs


def _ensemble(prefix, count):
    return [run for i in range(count) for run in default_runs("%s_%d" % (prefix, i + 1))]


CASES = {"HMG": case_HMG, "MAR": case_MAR, "REG_BAD": case_REG_BAD,
         "REG_ENS": lambda: _ensemble("1200_3_6_rand_ldpc", 10), "IREG_ENS": lambda: _ensemble("1200_rho_x5_rand_ldpc", 10)}


def lines(case, extra=(), all_decoders=False):
    out = []
    for run in CASES[case]():
        if all_decoders or run[2] in BUILT:
            out.append(" ".join(list(run) + list(extra)))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--all-decoders", action="store_true", help="also print the LP lines (that decoder is not built: those runs stop with an error)")
    ap.add_argument("case", nargs="+", choices=sorted(CASES), help="specify case(s)")
    ap.add_argument("arg", nargs=argparse.REMAINDER, help="arguments passed to wrapped command")
    a = ap.parse_args(argv)
    for case in a.case:
        for ln in lines(case, a.arg, a.all_decoders):
            print(ln, flush=True)


if __name__ == "__main__":
    main()
