"""ctypes binding of oracle/libmsm_oracle.so (plain-C restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libmsm_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    src = HERE / "msm_oracle.c"
    if force or not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
        res = subprocess.run(["make", "-C", str(HERE), "-B" if force else "-s"], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"oracle build failed:\n{res.stdout}\n{res.stderr}")
    return LIB


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB))
        _lib.oracle_count_transitions.restype = C.c_int64
        _lib.oracle_count_transitions.argtypes = [
            C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_state_counts.restype = None
        _lib.oracle_state_counts.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        _lib.oracle_kmeans_assign.restype = None
        _lib.oracle_kmeans_assign.argtypes = [
            C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_void_p]
        _lib.oracle_gemm_fma.restype = None
        _lib.oracle_gemm_fma.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.oracle_kmeans_fit.restype = C.c_int
        _lib.oracle_kmeans_fit.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_double,
                                           C.c_double, C.c_void_p, C.c_void_p]
    return _lib


def _bounds(segments, n):
    if segments is None:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    starts = np.ascontiguousarray([s for s, _ in segments], np.int64)
    stops = np.ascontiguousarray([e for _, e in segments], np.int64)
    return starts, stops


def count_transitions(labels, k, lag, *, segments=None, stride=1, weights=None):
    """-> (counts, n_pairs).  counts is int64 (unweighted) or float64 (weighted)."""
    lib = _load()
    labels = np.ascontiguousarray(labels, np.int32)
    starts, stops = _bounds(segments, labels.size)
    if weights is None:
        out = np.zeros((k, k), np.int64)
        pairs = lib.oracle_count_transitions(labels.ctypes.data, labels.size, starts.ctypes.data,
                                             stops.ctypes.data, len(starts), lag, stride, k, None,
                                             out.ctypes.data, None)
    else:
        w = np.ascontiguousarray(weights, np.float64)
        out = np.zeros((k, k), np.float64)
        pairs = lib.oracle_count_transitions(labels.ctypes.data, labels.size, starts.ctypes.data,
                                             stops.ctypes.data, len(starts), lag, stride, k,
                                             w.ctypes.data, None, out.ctypes.data)
    return out, int(pairs)


def state_counts(labels, k):
    lib = _load()
    labels = np.ascontiguousarray(labels, np.int32)
    out = np.zeros(k, np.int64)
    lib.oracle_state_counts(labels.ctypes.data, labels.size, k, out.ctypes.data)
    return out


def kmeans_assign(x, centers, mean=None, std=None, want_mindist=False):
    lib = _load()
    x = np.ascontiguousarray(x, np.float64)
    centers = np.ascontiguousarray(centers, np.float64)
    n, d = x.shape
    k = centers.shape[0]
    labels = np.empty(n, np.int32)
    md = np.empty(n, np.float64) if want_mindist else None
    m = np.ascontiguousarray(mean, np.float64) if mean is not None else None
    s = np.ascontiguousarray(std, np.float64) if std is not None else None
    lib.oracle_kmeans_assign(x.ctypes.data, n, d, d, centers.ctypes.data, k,
                             m.ctypes.data if m is not None else None,
                             s.ctypes.data if s is not None else None, labels.ctypes.data,
                             md.ctypes.data if md is not None else None)
    return (labels, md) if want_mindist else labels


def kmeans_fit(xz, k, seed=0, max_iter=50, tol2=0.0, n_total=None):
    """Engine's Lloyd fit restated on the CPU -> (centers, n_iter, scale).  xz already whitened."""
    lib = _load()
    xz = np.ascontiguousarray(xz, np.float64)
    n, d = xz.shape
    centers = np.empty((k, d), np.float64)
    scale = C.c_double()
    it = lib.oracle_kmeans_fit(xz.ctypes.data, n, d, k, int(seed), int(max_iter), float(tol2),
                               float(n if n_total is None else n_total), centers.ctypes.data, C.byref(scale))
    return centers, int(it), float(scale.value)


def gemm_fma(A, B):
    """A . B with every element the ascending-k fma chain from +0 (bit-level mirror of msm_gemm_f64)."""
    A = np.ascontiguousarray(A, np.float64)
    B = np.ascontiguousarray(B, np.float64)
    m, k = A.shape
    k2, n = B.shape
    assert k == k2
    Cm = np.empty((m, n), np.float64)
    _load().oracle_gemm_fma(A.ctypes.data, B.ctypes.data, Cm.ctypes.data, m, n, k)
    return Cm
