"""ctypes front-end of oracle/admm_oracle.c (ADMM LP decoder restated) and of oracle/_ref/libppolytope.so (the reference's own
projection.cpp, compiled from /root/reference by oracle/Makefile).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None
HARD_CAP = 100000


def build(force=False):
    so = os.path.join(_HERE, "libadmm_oracle.so")
    src = os.path.join(_HERE, "admm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libadmm_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.oracle_np_sum.restype = ctypes.c_double
    return _LIB


def ref_lib():
    """The reference's projection.cpp as built by `make -C oracle ref` (None when it has not been built)."""
    global _REF
    path = os.path.join(_HERE, "_ref", "libppolytope.so")
    if _REF is None and os.path.exists(path):
        _REF = ctypes.CDLL(path)
    return _REF


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def pp_project(v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros_like(v)
    rc = lib().oracle_pp_project(ctypes.c_int(v.size), _dp(v), _dp(out))
    if rc:
        raise ValueError("projection length %d unsupported" % v.size)
    return out


def pp_project_ref(v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros_like(v)
    ref_lib().proj_vec(ctypes.c_int(v.size), _dp(v), _dp(out))  # extern "C" proj_vec (projection.cpp:252)
    return out


def np_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().oracle_np_sum(_dp(a), ctypes.c_int64(a.size))


def admm_decode(g, gamma, mu, eps, max_iter):
    """g: object with m, n, chk, var.  -> (x float64 [B,n] as at return, iters int32 [B], converged uint8 [B])."""
    gamma = np.ascontiguousarray(np.atleast_2d(gamma), dtype=np.float64)
    B, n = gamma.shape
    chk = np.ascontiguousarray(g.chk, dtype=np.int32)
    var = np.ascontiguousarray(g.var, dtype=np.int32)
    x = np.zeros((B, n), dtype=np.float64)
    iters = np.zeros(B, dtype=np.int32)
    conv = np.zeros(B, dtype=np.uint8)
    rc = lib().oracle_admm_decode(ctypes.c_int(g.m), ctypes.c_int(g.n), ctypes.c_int64(len(chk)),
                                  chk.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), var.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                  _dp(gamma), ctypes.c_int64(B), ctypes.c_double(mu), ctypes.c_double(eps), ctypes.c_int(max_iter),
                                  ctypes.c_int(HARD_CAP), _dp(x), iters.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                  conv.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    if rc:
        raise RuntimeError("oracle_admm_decode rc=%d" % rc)
    return x, iters, conv


def pseudo_to_cw(x, allow_pseudo, eps=1e-8):
    """src/math_utils.py:28-34"""
    x = np.array(x, dtype=np.float64)
    if allow_pseudo:
        x[x < eps] = 0
        x[1 - x < eps] = 1
        return x
    return (x > .5).astype(int)
