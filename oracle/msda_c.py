"""TEST INFRASTRUCTURE ONLY - ctypes wrapper over oracle/msda_ref.c (scalar restatement of the CUDA kernel)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmsda_ref.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "msda_ref.c")):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def msda_forward(value, shapes, loc, aw):
    """numpy: value (B,S,M,D), shapes (L,2) int64, loc (B,Q,M,L,P,2), aw (B,Q,M,L,P) -> (B,Q,M*D)."""
    lib = _load()
    dt = value.dtype
    assert dt in (np.float32, np.float64)
    value, loc, aw = (np.ascontiguousarray(a, dtype=dt) for a in (value, loc, aw))
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.concatenate([[0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1]]).astype(np.int64)
    b, s, m, d = value.shape
    _, q, _, l, p, _ = loc.shape
    out = np.empty((b, q, m * d), dtype=dt)
    fn = lib.msda_ref_f64 if dt == np.float64 else lib.msda_ref_f32
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = fn(ptr(value), ptr(shapes), ptr(lsi), ptr(loc), ptr(aw), ptr(out), b, s, m, d, l, q, p)
    assert rc == 0
    return out


def msda_backward(value, shapes, loc, aw, grad_out):
    """numpy -> (grad_value (B,S,M,D), grad_loc (B,Q,M,L,P,2), grad_aw (B,Q,M,L,P))."""
    lib = _load()
    dt = value.dtype
    assert dt in (np.float32, np.float64)
    value, loc, aw, grad_out = (np.ascontiguousarray(a, dtype=dt) for a in (value, loc, aw, grad_out))
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.concatenate([[0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1]]).astype(np.int64)
    b, s, m, d = value.shape
    _, q, _, l, p, _ = loc.shape
    gv, gl, ga = np.empty_like(value), np.empty_like(loc), np.empty_like(aw)
    fn = lib.msda_ref_bwd_f64 if dt == np.float64 else lib.msda_ref_bwd_f32
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = fn(ptr(value), ptr(shapes), ptr(lsi), ptr(loc), ptr(aw), ptr(grad_out), ptr(gv), ptr(gl), ptr(ga), b, s, m, d, l, q, p)
    assert rc == 0
    return gv, gl, ga
