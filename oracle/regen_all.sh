#!/bin/bash
# ONE command that rebuilds every fixture under tests/golden/ (and the shipped H files) from the reference -- build container only
# (/root/reference must exist; nothing here runs on the GPU box).  Each file has exactly one writer:
#   make_goldens.py         codes_edges.npz, kat.json, decode_*.npz, main_counters.json
#   make_goldens_ml.py      ml_vectors.npz, ml_cases.json, ml_kat.json, main_counters_ml.json
#   make_goldens_admm.py    admm_vectors.npz, admm_cases.json, main_counters_admm.json   (needs oracle/_ref/libppolytope.so: make -C oracle)
#   make_goldens_gen.py     irregular_ensembles.json
#   make_goldens_curves.py  published_curves.json, reference_checks.json
#   make_goldens_codes.py   ldpc_decoders_amd/data/codes/*.txt
#   make_timing.py          reference_timing.json  (about five minutes of single-core timing; skipped with --no-timing)
# Afterwards `git status tests/golden` must be clean: the fixtures are a pure function of the reference and these scripts
# (reference_timing.json excepted -- it is a measurement).   Usage: oracle/regen_all.sh [--no-timing]
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
cd "$HERE/.."
test -d /root/reference || { echo "regen_all.sh: /root/reference is not here (build container only)"; exit 1; }
make -s -C oracle all
python oracle/make_goldens_codes.py
python oracle/make_goldens.py
python oracle/make_goldens_ml.py
python oracle/make_goldens_admm.py
python oracle/make_goldens_gen.py
python oracle/make_goldens_curves.py
if [ "${1:-}" != "--no-timing" ]; then python oracle/make_timing.py; fi
git status --short tests/golden ldpc_decoders_amd/data/codes
