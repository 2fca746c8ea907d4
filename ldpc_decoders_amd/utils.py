"""CLI flags, logging and the JSON result store -- mirror of the reference's ``src/utils.py:21-68,118-140``.

The argument grammar (positional ``channel code decoder`` + ``--codeword --min-wec --params --max-iter ...``) and
the result files (``<channel>-<code>-<decoder>-<codeword>-<min_wec>-<max_iter>.json`` with the id keys first, then
``tot wec wer bec ber`` as ``{str(param): value}``) are what ``simulations.py`` / ``run_sims.sh`` emit and what
``graph.py`` reads upstream (src/graph.py:25-58), so they are kept verbatim.  Flags that only concern decoders
outside the BP path (``--mu --eps --allow-pseudo --layers --train --apprx``) are accepted and ignored.
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


def setup_parser(code_names, channel_names, decoder_names):
    p = argparse.ArgumentParser()
    p.add_argument("channel", help="channel type", choices=list(channel_names))
    p.add_argument("code", help="code name; built-ins plus the files in $%s (default data/codes)" % codes.file_codes_dir_string,
                   choices=list(code_names))
    p.add_argument("decoder", help="decoder type", choices=list(decoder_names))
    p.add_argument("--codeword", help="transmitted codeword [0:all-zero, 1:all-ones, -1:random from code book (small codes, host noise)]",
                   default=0, type=int, choices=[-1, 0, 1])
    p.add_argument("--min-wec", help="min word errors to accumulate", default=100, type=int)
    p.add_argument("--params", help="channel condition, e.g. erasure probability for erasure channel", nargs="+", type=float,
                   default=[.1, .01])
    p.add_argument("--max-iter", help="max iteration count for iterative decoders", default=10, type=int)
    # accepted for arg-line compatibility with simulations.py; unused by SPA/MSA
    p.add_argument("--mu", default=3., type=float, help=argparse.SUPPRESS)
    p.add_argument("--eps", default=1e-5, type=float, help=argparse.SUPPRESS)
    p.add_argument("--allow-pseudo", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--layers", nargs="+", default=[100, 100], type=int, help=argparse.SUPPRESS)
    p.add_argument("--train", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--apprx", default=-1, type=int, help=argparse.SUPPRESS)
    p.add_argument("--log-freq", help="log frequency in seconds", default=5., type=float)
    # GPU build additions
    p.add_argument("--precision", choices=["f32", "f64"], default=None,
                   help="message arithmetic (default: f32 with device noise; f64 with --exact and for min-sum over the BSC, which is tie-dominated)")
    p.add_argument("--backend", choices=["auto", "stream", "fused"], default="auto", help="kernel family")
    p.add_argument("--batch", type=int, default=65536, help="frames per round and per GPU (device-noise mode)")
    p.add_argument("--seed", type=int, default=0x5EED1200, help="Philox seed of the device noise")
    p.add_argument("--exact", action="store_true",
                   help="reference-exact mode: host numpy noise (np.random global stream), fp64 messages, sequential stopping rule")
    p.add_argument("--np-seed", type=int, default=None, help="np.random.seed() for --exact runs (upstream runs unseeded)")
    return bind_parser_common(p)


def bind_parser_common(parser):
    _dir = default_data_root()
    path_ = lambda p_: os.path.abspath(os.path.join(_dir, p_))  # noqa: E731
    parser.add_argument("--data_dir", help="location for writing simulation output", default=path_("data"))
    parser.add_argument("--cache_dir", help="unused (ADMMA cache upstream)", default=path_("cache"))
    parser.add_argument("--plots_dir", help="save location of plots", default=path_("plots"))
    parser.add_argument("--debug", help="logs debug info", action="store_true")
    parser.add_argument("--console", help="if true prints log onto console, otherwise write to a file", action="store_true")
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
