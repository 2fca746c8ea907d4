"""Import the upstream reference (thadikari/ldpc_decoders) in THIS container only.

TEST INFRASTRUCTURE -- never imported by the product package.  Used by
``oracle/make_goldens.py`` to generate the committed fixtures under
``tests/golden/`` and -- through that script -- by ``tests/test_goldens_regenerate_cpu.py`` (skipped when
``/root/reference`` is absent, e.g. on the GPU box).

The reference needs two in-memory shims; nothing is
written into the reference tree:
  * a stub ``utilities`` package (``src/utilities`` is an empty, un-vendored git
    submodule; used at reference ``src/utils.py:11-12,48``);
  * ``np.int`` / ``np.NINF`` aliases removed from numpy>=1.24/2.0
    (``src/math_utils.py:25``, ``src/bec.py:34``).
"""
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("LDPC_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "src"))


def load(tmp_dir="/tmp/ldpc_oracle_ref"):
    """Return a namespace with the reference modules (utils, codes, bpa, biawgn, bsc, bec)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "NINF"):
        np.NINF = -np.inf
    if "utilities" not in sys.modules:
        ut = types.ModuleType("utilities")
        ut.__path__ = []
        f = types.ModuleType("utilities.file")
        f.resolve_data_dir_os = lambda name: os.path.join(tmp_dir, name)

        class Registry(dict):
            def put(self, k, v):
                self[k] = v

            def reg(self, fn):
                self[fn.__name__] = fn
                return fn

        ut.file, ut.Registry = f, Registry
        sys.modules["utilities"], sys.modules["utilities.file"] = ut, f
    os.environ.setdefault("FILE_CODES_DIR", os.path.join(REF_ROOT, "data", "codes"))
    src = os.path.join(REF_ROOT, "src")
    if src not in sys.path:
        sys.path.insert(0, src)
    import importlib

    ns = types.SimpleNamespace()
    for name in ("utils", "codes", "math_utils", "bpa", "biawgn", "bsc", "bec", "models"):
        setattr(ns, name, importlib.import_module(name))
    return ns
