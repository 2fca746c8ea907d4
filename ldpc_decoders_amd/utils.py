"""CLI flags, logging and the JSON result store -- mirror of the reference's ``src/utils.py:21-68,118-140``.

The argument grammar (positional ``channel code decoder`` + ``--codeword --min-wec --params --max-iter ...``) and
the result files (``<channel>-<code>-<decoder>-<codeword>-<min_wec>-<max_iter>.json`` with the id keys first, then
``tot wec wer bec ber`` as ``{str(param): value}``) are what ``simulations.py`` / ``run_sims.sh`` emit and what
``graph.py`` reads upstream (src/graph.py:25-58), so flag names, defaults and the file layout are kept; the help texts are
this build's own.  ``--mu --eps --allow-pseudo`` configure the ADMM decoder; ``--layers --train --apprx`` belong to ADMMA, which
is not built -- they are accepted so that upstream arg-lines parse, and ignored.
"""
import argparse
import json
import logging
import os
from collections import OrderedDict

from . import codes
from .models import decoder_names  # noqa: F401  (re-exported like upstream utils.decoder_names)

strl = lambda ll: (str(it_) for it_ in ll)  # noqa: E731


def default_data_root():
    # upstream resolves this through its un-vendored `utilities` submodule (src/utils.py:48); here: env or ~/decoders
    return os.environ.get("LDPC_DATA_ROOT", os.path.join(os.path.expanduser("~"), "decoders"))


def _reference_flags(p, channel_names, code_names, decoder_names):
    """The reference's flag grammar (names, types, defaults, choices: src/utils.py:21-45) -- the contract simulations.py's arg-lines
    rely on.  Help texts are this build's own."""
    p.add_argument("channel", choices=list(channel_names), help="which channel model to simulate")
    p.add_argument("code", choices=list(code_names),
                   help="name of the code: a built-in, or a parity-check text file found in $%s (falls back to data/codes)" % codes.file_codes_dir_string)
    p.add_argument("decoder", choices=list(decoder_names), help="which decoder to run on the GPU")
    p.add_argument("--codeword", type=int, default=0, choices=[-1, 0, 1],
                   help="what is sent: 0 = the all-zero word, 1 = the all-one word, -1 = a fresh random codeword per frame (codes with a code book, i.e. short ones)")
    p.add_argument("--min-wec", type=int, default=100, help="stop a point once this many frames were decoded wrongly")
    p.add_argument("--params", type=float, nargs="+", default=[.1, .01],
                   help="one run per value: SNR in dB (biawgn), crossover probability (bsc) or erasure probability (bec)")
    p.add_argument("--max-iter", type=int, default=10, help="sweep cap of the iterative decoders (<= 0: no cap)")
    p.add_argument("--mu", type=float, default=3., help="ADMM penalty parameter")
    p.add_argument("--eps", type=float, default=1e-5, help="ADMM stopping tolerance")
    p.add_argument("--allow-pseudo", action="store_true", help="ADMM: keep fractional (pseudo-codeword) outputs instead of rounding them")
    # decoders that are not built here (ADMMA): accepted so that upstream arg-lines parse, otherwise unused
    p.add_argument("--layers", type=int, nargs="+", default=[100, 100], help=argparse.SUPPRESS)
    p.add_argument("--train", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--apprx", type=int, default=-1, help=argparse.SUPPRESS)
    p.add_argument("--log-freq", type=float, default=5., help="seconds between progress lines / intermediate result writes")


def setup_parser(code_names, channel_names, decoder_names):
    p = argparse.ArgumentParser()
    _reference_flags(p, channel_names, code_names, decoder_names)
    g = p.add_argument_group("GPU build")
    g.add_argument("--precision", choices=["f32", "f64", "f16"], default=None,
                   help="message arithmetic (default: f32 with device noise; f64 with --exact and for min-sum over the BSC, which is tie-dominated); "
                        "f16: fp16 STORAGE of the messages on the streaming kernels, fp32 arithmetic -- a tolerance mode for codes whose state lives in HBM")
    g.add_argument("--backend", choices=["auto", "stream", "fused"], default="auto", help="kernel family")
    g.add_argument("--batch", type=int, default=65536, help="frames per round and per GPU (device-noise mode)")
    g.add_argument("--seed", type=int, default=0x5EED1200, help="Philox seed of the device noise")
    g.add_argument("--max-frames", type=int, default=0,
                   help="device-noise mode: stop a parameter after this many frames even if --min-wec was not reached (0 = no cap, as upstream)")
    g.add_argument("--exact", action="store_true",
                   help="reference-exact mode: host numpy noise (np.random global stream), fp64 messages, sequential stopping rule")
    g.add_argument("--np-seed", type=int, default=None, help="np.random.seed() for --exact runs (upstream runs unseeded)")
    g.add_argument("--prior-grid", type=int, default=None, metavar="K",
                   help="exact-in-fp32 min-sum over BI-AWGN: LLRs rounded to multiples of 2^-K, decoded by the fp32 LDS kernels under an exactness "
                        "guard; the few frames beyond it are decoded again in fp64 -- every counted frame is what the fp64 reference returns "
                        "for those priors")
    return bind_parser_common(p)


def bind_parser_common(parser):
    root = default_data_root()
    for flag, sub, text in (("--data_dir", "data", "directory the JSON result files go to"),
                            ("--cache_dir", "cache", "accepted for upstream compatibility (ADMMA cache), unused"),
                            ("--plots_dir", "plots", "directory for figures")):
        parser.add_argument(flag, default=os.path.abspath(os.path.join(root, sub)), help=text)
    parser.add_argument("--debug", action="store_true", help="log at DEBUG level")
    parser.add_argument("--console", action="store_true", help="log to the terminal instead of <data_dir>/test.log")
    return parser


def setup_console_logger(level=logging.DEBUG):
    logging.basicConfig(format="%(name)s|%(message)s", level=level)


def setup_file_logger(path, name, level=logging.DEBUG):
    logging.basicConfig(filename=os.path.join(path, "%s.log" % name), filemode="a",
                        format="%(asctime)s,%(msecs)03d|%(name)s|%(levelname)s|%(message)s", datefmt="%H:%M:%S", level=level)
    logging.info("Logger init to file. %s" % ("%" * 80))


def make_dir_if_not_exists(dir_path):
    os.makedirs(dir_path, exist_ok=True)


def load_json(file_path):
    try:
        with open(file_path, "r") as ff:
            return json.load(ff, object_pairs_hook=OrderedDict)
    except (OSError, ValueError):
        return None


class Saver:
    """Read-modify-write JSON store, one file per run id (src/utils.py:118-140)."""

    def __init__(self, data_dir, run_ids):
        self.dict = OrderedDict(run_ids)
        make_dir_if_not_exists(data_dir)
        self.file_path = os.path.join(data_dir, "%s.json" % "-".join(strl(self.dict.values())))

    def add_meta(self, key, val):
        self.dict[key] = val

    def add(self, param, val_dict):
        data = load_json(self.file_path)
        if data is None:
            data = OrderedDict()
            for key in self.dict:
                data[key] = self.dict[key]
            for key in val_dict:
                data[key] = {}
        for key in val_dict:
            data.setdefault(key, {})[str(param)] = val_dict[key]
        with open(self.file_path, "w") as fp:
            json.dump(data, fp, indent=4)
