"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/local_attn_ref.c."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_local_attn.so")


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def _lib():
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    fp = ctypes.POINTER(ctypes.c_float)
    for name in ("oracle_local_similar", "oracle_local_weighting"):
        getattr(lib, name).argtypes = [fp, fp, fp] + [ctypes.c_int] * 6
        getattr(lib, name).restype = None
    for name in ("oracle_local_similar_mt", "oracle_local_weighting_mt"):
        getattr(lib, name).argtypes = [fp, fp, fp] + [ctypes.c_int] * 7
        getattr(lib, name).restype = None
    lib.oracle_max_threads.restype = ctypes.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def local_similar(q: np.ndarray, k: np.ndarray, kH: int, kW: int) -> np.ndarray:
    q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float32)
    N, C, H, W = q.shape
    s = np.empty((N, H, W, kH * kW), np.float32)
    _lib().oracle_local_similar(_p(q), _p(k), _p(s), N, C, H, W, kH, kW)
    return s


def local_weighting(v: np.ndarray, w: np.ndarray, kH: int, kW: int) -> np.ndarray:
    v = np.ascontiguousarray(v, np.float32); w = np.ascontiguousarray(w, np.float32)
    N, C, H, W = v.shape
    o = np.empty_like(v)
    _lib().oracle_local_weighting(_p(v), _p(w), _p(o), N, C, H, W, kH, kW)
    return o


def local_similar_mt(q: np.ndarray, k: np.ndarray, kH: int, kW: int, nthreads: int = 0) -> np.ndarray:
    """oracle_local_similar_mt: the same function, image rows in parallel over `nthreads` OpenMP threads (0: the runtime's default)."""
    q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float32)
    N, C, H, W = q.shape
    s = np.empty((N, H, W, kH * kW), np.float32)
    _lib().oracle_local_similar_mt(_p(q), _p(k), _p(s), N, C, H, W, kH, kW, nthreads)
    return s


def local_weighting_mt(v: np.ndarray, w: np.ndarray, kH: int, kW: int, nthreads: int = 0) -> np.ndarray:
    v = np.ascontiguousarray(v, np.float32); w = np.ascontiguousarray(w, np.float32)
    N, C, H, W = v.shape
    o = np.empty_like(v)
    _lib().oracle_local_weighting_mt(_p(v), _p(w), _p(o), N, C, H, W, kH, kW, nthreads)
    return o
