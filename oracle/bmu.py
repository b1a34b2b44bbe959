"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front end of oracle/bmu_oracle.c
(the plain-C restatement of Codebook.get_patches_bmu, reference
models/Codebook.py:77-99)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libqarig_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "bmu_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_get_threads.restype = ctypes.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def set_threads(n):
    lib().oracle_set_threads(int(n))


def get_threads():
    return int(lib().oracle_get_threads())


def patchify(x, patch_dim):
    x = _f32(x)
    N, C, H, W = x.shape
    pH, pW = patch_dim
    out = np.empty((N * (H // pH) * (W // pW), C * pH * pW), dtype=np.float32)
    lib().oracle_patchify(_p(x), N, C, H, W, pH, pW, _p(out))
    return out


def bmu(x, codebook, patch_dim):
    """int64 (N*Seq,) indices; x (N,C,H,W), codebook (K,D) array-likes."""
    x = _f32(x)
    w = _f32(codebook)
    N, C, H, W = x.shape
    pH, pW = patch_dim
    K, D = w.shape
    assert D == C * pH * pW
    out = np.empty(N * (H // pH) * (W // pW), dtype=np.int64)
    lib().oracle_bmu(_p(x), N, C, H, W, pH, pW, _p(w), K, _p(out))
    return out


def bmu_f64(x, codebook, patch_dim):
    """(argmin in double, top-2 gap in double) per patch row."""
    x = _f32(x)
    w = _f32(codebook)
    N, C, H, W = x.shape
    pH, pW = patch_dim
    K, D = w.shape
    R = N * (H // pH) * (W // pW)
    idx = np.empty(R, dtype=np.int64)
    gap = np.empty(R, dtype=np.float64)
    lib().oracle_bmu_f64(_p(x), N, C, H, W, pH, pW, _p(w), K, _p(idx), _p(gap))
    return idx, gap
