"""ctypes binding of libldpc_hip.so (C ABI: include/ldpc_hip.h).

Mirrors the upstream pattern for native code (``src/parity_polytope/exact.py:12-21``: lazy
``ctypes.cdll.LoadLibrary`` + typed argument lists).  There is deliberately NO fallback: if the HIP
library is missing or no GPU is usable, the product path raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LDPC_LIB_PATH") or os.path.join(_HERE, "csrc", "libldpc_hip.so")  # override: A/B builds of the same ABI

ALG = {"MSA": 0, "SPA": 1, "BEC": 2}
DTYPE = {"f32": 0, "f64": 1, "f16": 2}  # f16: fp16 storage of the streaming messages, fp32 arithmetic / priors (decoders only)
IO_DTYPE = {"f32": 0, "f64": 1, "f16": 0}  # what the channel kernels write / the decoders read for each decoder precision
BACKEND = {"auto": 0, "stream": 1, "fused": 2}
BACKEND_NAME = {v: k for k, v in BACKEND.items()}
CHANNEL = {"biawgn": 0, "bsc": 1, "bec": 2}
CH_RAW_OBSERVATION = 0x100
FLAG_NO_EARLY_EXIT = 1


def flag_prior_grid(k):
    """LDPC_FLAG_PRIOR_GRID(k) of include/ldpc_hip.h: exact-in-fp32 min-sum on priors rounded to multiples of 2^-k (None: off)."""
    return 0 if k is None else (int(k) + 1) << 8


def ch_prior_grid(k):
    """LDPC_CH_PRIOR_GRID(k): the channel kernel rounds the LLRs it writes to multiples of 2^-k (None: off)."""
    return 0 if k is None else (int(k) + 1) << 12


CNT_TOT, CNT_WEC, CNT_BEC, CNT_ITER_SUM, CNT_HIST0 = 0, 1, 2, 3, 4

_c = ctypes
_P = ctypes.c_void_p
SIGNATURES = {
    "ldpc_last_error": (_c.c_char_p, []),
    "ldpc_abi_version": (_c.c_int, []),
    "ldpc_device_count": (_c.c_int, [_c.POINTER(_c.c_int)]),
    "ldpc_code_create": (_c.c_int, [_c.c_int, _c.c_int32, _c.c_int32, _c.c_int64, _P, _P, _c.POINTER(_P)]),
    "ldpc_code_destroy": (_c.c_int, [_P]),
    "ldpc_code_info": (_c.c_int, [_P, _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int64),
                                  _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32)]),
    "ldpc_decoder_create": (_c.c_int, [_P, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_P)]),
    "ldpc_decoder_destroy": (_c.c_int, [_P]),
    "ldpc_decoder_last_stats": (_c.c_int, [_P, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "ldpc_decoder_last_repacks": (_c.c_int, [_P, _c.POINTER(_c.c_int)]),
    "ldpc_decoder_chunk_state": (_c.c_int, [_P, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int)]),
    "ldpc_decoder_grid_violations": (_c.c_int, [_P, _c.POINTER(_c.c_int64), _P, _c.c_int64, _c.c_int]),
    "ldpc_decoder_grid_list": (_c.c_int, [_P, _c.POINTER(_P), _c.POINTER(_c.c_int64), _P]),
    "ldpc_decoder_grid_list_reset": (_c.c_int, [_P, _P]),
    "ldpc_channel_list": (_c.c_int, [_c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_uint64, _c.c_uint64, _P, _c.c_int64, _c.c_int32, _P, _P]),
    "ldpc_count_errors_list": (_c.c_int, [_P, _c.c_int, _P, _P, _c.c_int64, _c.c_int32, _c.c_int32, _P, _c.c_int64, _c.c_uint64, _c.c_uint64, _c.c_int64, _P, _P]),
    "ldpc_decoder_fused_info": (_c.c_int, [_P, _c.POINTER(_c.c_double)]),
    "ldpc_plan_layout": (_c.c_int, [_c.c_int32, _c.c_int32, _c.c_int64, _P, _P, _c.c_int, _c.c_int, _c.c_int64, _c.c_char_p,
                                    _c.POINTER(_c.c_double)]),
    "ldpc_decoder_profile": (_c.c_int, [_P, _c.c_int]),
    "ldpc_decoder_profile_read": (_c.c_int, [_P, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64), _c.c_int]),
    "ldpc_decoder_kernel_name": (_c.c_int, [_P, _c.c_int, _c.c_char_p, _c.c_int64]),
    "ldpc_decode": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int32, _c.c_uint32, _P, _P, _P]),
    "ldpc_decode_bits": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int32, _c.c_uint32, _P, _P, _P, _P]),
    "ldpc_decode_host_bits": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int32, _c.c_uint32, _P, _P, _P]),
    "ldpc_count_errors_bits": (_c.c_int, [_P, _P, _P, _c.c_int, _P, _c.c_int64, _c.c_int32, _c.c_int32, _P, _P]),
    "ldpc_decode_soft": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int32, _c.c_uint32, _P, _P, _P, _P]),
    "ldpc_decode_host": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int32, _c.c_uint32, _P, _P]),
    "ldpc_channel": (_c.c_int, [_c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_uint64, _c.c_uint64, _c.c_uint64, _c.c_int64,
                                _c.c_int32, _P, _P, _P]),
    "ldpc_channel_words": (_c.c_int, [_c.c_int, _c.c_int, _c.c_double, _P, _c.c_int64, _c.c_uint64, _c.c_uint64, _c.c_uint64, _c.c_int64,
                                      _c.c_int32, _P, _P, _P, _P]),
    "ldpc_count_errors_words": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int32, _c.c_int32, _P, _P]),
    "ldpc_count_errors": (_c.c_int, [_P, _P, _c.c_int, _P, _c.c_int64, _c.c_int32, _c.c_int32, _P, _P]),
    "ldpc_debug_copy4": (_c.c_int, [_P, _P, _c.c_int64, _P]),
    "ldpc_simulate": (_c.c_int, [_P, _c.c_int, _c.c_double, _c.c_int, _c.c_uint64, _c.c_uint64, _c.c_uint64, _c.c_int64,
                                 _c.c_int32, _c.c_uint32, _c.c_int32, _P, _P]),
    "ldpc_simulate_rounds": (_c.c_int, [_P, _c.c_int, _c.c_double, _c.c_int, _c.c_uint64, _c.c_uint64, _c.c_uint64, _c.c_int64, _c.c_int32,
                                        _c.c_uint64, _c.c_int32, _c.c_uint32, _c.c_int32, _P, _P]),
    "ldpc_ml_create": (_c.c_int, [_c.c_int, _P, _c.c_int64, _c.c_int32, _c.POINTER(_P)]),
    "ldpc_ml_destroy": (_c.c_int, [_P]),
    "ldpc_ml_decode": (_c.c_int, [_P, _c.c_int, _c.c_int, _c.POINTER(_c.c_double), _P, _c.c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "ldpc_admm_create": (_c.c_int, [_P, _c.POINTER(_P)]),
    "ldpc_admm_destroy": (_c.c_int, [_P]),
    "ldpc_admm_last_repacks": (_c.c_int, [_P, _c.POINTER(_c.c_int)]),
    "ldpc_admm_last_backend": (_c.c_int, [_P, _c.POINTER(_c.c_int)]),
    "ldpc_admm_decode": (_c.c_int, [_P, _P, _c.c_int64, _c.c_double, _c.c_double, _c.c_int32, _P, _P, _P, _P]),
    "ldpc_ml_simulate": (_c.c_int, [_P, _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_uint64, _c.c_uint64, _c.c_uint64,
                                    _c.c_int64, _P, _P]),
}

_lib = None


class LdpcHipError(RuntimeError):
    pass


def library_path():
    """Path of the shared library `load()` opens (bench.py hashes its kernels against the committed PMC counters)."""
    return LIB_PATH


def load():
    """Load the HIP library (once).  Raises if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LdpcHipError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(or `make -C ldpc_decoders_amd/csrc`); there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7 and owns the streams / device
        # memory this library is handed.  Importing torch first makes the dynamic loader bind libldpc_hip.so to that
        # already-loaded runtime (same soname) instead of a second copy from /opt/rocm.
        import torch  # noqa: F401

        lib = ctypes.CDLL(LIB_PATH)
        older_build = os.environ.get("LDPC_LIB_ALLOW_OLDER_ABI") == "1"  # A/B runs against a library of an earlier round (tools/ab_sim.sh)
        for name, (res, args) in SIGNATURES.items():
            if older_build and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)  # AttributeError here == ABI mismatch with include/ldpc_hip.h
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().ldpc_last_error()
        raise LdpcHipError("libldpc_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def device_count():
    n = ctypes.c_int(0)
    check(load().ldpc_device_count(ctypes.byref(n)))
    return n.value
