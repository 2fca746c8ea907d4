#!/usr/bin/env python3
"""Collect the reference's PUBLISHED result files (data/output/*.json: BER/WER curves it ships) for every code the build can
construct (all of its data/codes files, kept as fixtures, and the built-in toy codes) into tests/golden/published_curves.json.
Data only (counters per channel parameter)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

# every code whose H the build can construct: all 27 data/codes files (shipped in ldpc_decoders_amd/data/codes) + the built-in toy codes
CODES = {os.path.splitext(f)[0] for f in os.listdir(os.path.join(ROOT, "ldpc_decoders_amd", "data", "codes"))} | {"4_2_test", "6_2_3_ldpc", "7_4_hamming", "12_3_4_ldpc"}
DECODERS = {"SPA", "MSA", "ML", "ADMM"}

out = []
src = os.path.join(ref_import.REF_ROOT, "data", "output")
for name in sorted(os.listdir(src)):
    if not name.endswith(".json"):
        continue
    with open(os.path.join(src, name)) as fp:
        d = json.load(fp)
    if d.get("code") not in CODES or d.get("decoder") not in DECODERS or "ber" not in d:
        continue
    keys = [k for k in d if k not in ("tot", "wec", "wer", "bec", "ber", "dec")]
    rec = {k: d[k] for k in keys}
    rec["file"] = name
    rec["points"] = {p: dict(tot=d["tot"][p], wec=d["wec"][p], wer=d["wer"][p], bec=d["bec"][p], ber=d["ber"][p]) for p in d["ber"]}
    out.append(rec)
with open(os.path.join(ROOT, "tests", "golden", "published_curves.json"), "w") as fp:
    json.dump(out, fp, indent=0)
print(len(out), "curves,", sum(len(r["points"]) for r in out), "points")
for r in out:
    print(" ", r["file"], {k: v for k, v in r.items() if k not in ("points", "file")}, len(r["points"]))


def reference_checks():
    """Two facts about the CURRENT upstream code that decide how its published files can be used (bounded runs, ~2 minutes):
    (1) its sum-product decoder is not codeword-symmetric -- NaN / saturation artefacts decode towards the all-zero word -- so
        its all-zero-codeword curves at many iterations are optimistic; (2) some older-format published files (no `codeword`
        key) are not reproduced by the current code itself."""
    import numpy as np

    R = ref_import.load()
    code = R.codes.get_code("1200_3_6_ldpc")
    out = {"spa_codeword_asymmetry": [], "bsc_msa_current_vs_published": []}
    for ch, p, mi, n in (("bsc", 0.07, 40, 60), ("bsc", 0.06, 10, 60), ("biawgn", 1.75, 40, 60)):
        mod = getattr(R, ch)
        rec = dict(channel=ch, param=p, max_iter=mi, frames=n)
        for cw in (0, 1):
            np.random.seed(7)
            chan, dec = mod.Channel(p), mod.SPA(p, code, max_iter=mi)
            x = code.parity_mtx[0] * 0 + cw
            be = we = 0
            with np.errstate(all="ignore"):
                for _ in range(n):
                    e = int((dec.decode(chan.send(x)) != x).sum())
                    be, we = be + e, we + (e > 0)
            rec["codeword%d" % cw] = dict(wer=we / n, ber=be / (n * 1200))
        out["spa_codeword_asymmetry"].append(rec)
        print("  SPA asymmetry:", rec, flush=True)
    pub = {c["file"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "published_curves.json")))}
    for mi in (1, 2, 3, 6, 10, 40):
        p = "0.0451"
        np.random.seed(3)
        chan, dec = R.bsc.Channel(float(p)), R.bsc.MSA(float(p), code, max_iter=mi)
        x = code.parity_mtx[0] * 0
        be = we = 0
        n = 60
        for _ in range(n):
            e = int((dec.decode(chan.send(x)) != x).sum())
            be, we = be + e, we + (e > 0)
        pp = pub["bsc-1200_3_6_ldpc-MSA-%d.json" % mi]["points"][p]
        rec = dict(max_iter=mi, param=float(p), frames=n, current=dict(wer=we / n, ber=be / (n * 1200)), published=dict(wer=pp["wer"], ber=pp["ber"]))
        out["bsc_msa_current_vs_published"].append(rec)
        print("  BSC MSA:", rec, flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "reference_checks.json"), "w") as fp:
        json.dump(out, fp, indent=1)


if __name__ == "__main__" and "--checks" in sys.argv:
    reference_checks()
