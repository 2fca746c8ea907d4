"""Copies the reference's parity-check DATA files (data/codes/*.txt: one check per line, whitespace-separated variable numbers) into
ldpc_decoders_amd/data/codes/ -- package data, so that every code name of the reference's experiment tables (simulations.py: HMG, MAR, REG_ENS,
IREG_ENS) resolves on a machine without the reference, and the shipped layout plans can be tested against the matrices they were
made for.  Build container only:  python oracle/make_goldens_codes.py"""
import os
import shutil

SRC = "/root/reference/data/codes"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ldpc_decoders_amd", "data", "codes")

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    for f in sorted(os.listdir(SRC)):
        if f.endswith(".txt"):
            shutil.copyfile(os.path.join(SRC, f), os.path.join(DST, f))
    print(len(os.listdir(DST)), "code files in", DST)
