"""ctypes front-end of the plain-C oracle (oracle/bp_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libbp_oracle.so")
    src = os.path.join(_HERE, "bp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libbp_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def bp_decode(g, alg, y0, priors, max_iter, dtype=np.float64, nthreads=0):
    """g: any object with m, n, chk, var (row-major edges).  Returns (xhat uint8 [B,n], iters int32 [B])."""
    L = lib()
    priors = np.ascontiguousarray(np.atleast_2d(priors), dtype=dtype)
    B, n = priors.shape
    y0a = None if y0 is None else np.ascontiguousarray(np.atleast_2d(y0), dtype=dtype)
    chk = np.ascontiguousarray(g.chk, dtype=np.int32)
    var = np.ascontiguousarray(g.var, dtype=np.int32)
    xhat = np.zeros((B, n), dtype=np.uint8)
    iters = np.zeros(B, dtype=np.int32)
    ct = ctypes.c_double if dtype == np.float64 else ctypes.c_float
    fn = L.oracle_bp_decode_f64 if dtype == np.float64 else L.oracle_bp_decode_f32
    fn.restype = ctypes.c_int
    rc = fn(ctypes.c_int(g.m), ctypes.c_int(g.n), ctypes.c_int64(len(chk)), _p(chk, ctypes.c_int32), _p(var, ctypes.c_int32),
            ctypes.c_int(0 if alg == "MSA" else 1), _p(y0a, ct), _p(priors, ct), ctypes.c_int64(B), ctypes.c_int(max_iter),
            _p(xhat, ctypes.c_uint8), _p(iters, ctypes.c_int32), ctypes.c_int(nthreads))
    if rc:
        raise RuntimeError("oracle_bp_decode rc=%d" % rc)
    return xhat, iters


def bec_decode(g, y, max_iter, nthreads=0):
    L = lib()
    y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.uint8)
    B, n = y.shape
    chk = np.ascontiguousarray(g.chk, dtype=np.int32)
    var = np.ascontiguousarray(g.var, dtype=np.int32)
    xhat = np.zeros((B, n), dtype=np.uint8)
    iters = np.zeros(B, dtype=np.int32)
    L.oracle_bec_decode.restype = ctypes.c_int
    rc = L.oracle_bec_decode(ctypes.c_int(g.m), ctypes.c_int(g.n), ctypes.c_int64(len(chk)), _p(chk, ctypes.c_int32),
                             _p(var, ctypes.c_int32), _p(y, ctypes.c_uint8), ctypes.c_int64(B), ctypes.c_int(max_iter),
                             _p(xhat, ctypes.c_uint8), _p(iters, ctypes.c_int32), ctypes.c_int(nthreads))
    if rc:
        raise RuntimeError("oracle_bec_decode rc=%d" % rc)
    return xhat, iters


def num_threads():
    return lib().oracle_num_threads()
