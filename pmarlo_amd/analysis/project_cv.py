"""Learned output whitening of projected CVs: mirror of pmarlo.analysis.project_cv.apply_whitening_from_metadata
(S/analysis/project_cv.py:15-61) and pmarlo.ml.deeptica.whitening.apply_output_transform / _coerce_bool_flag
(S/ml/deeptica/whitening.py:10-175).

The transform is affine in every stage: (x - mean) W, minus the batch drift, times the inverse Cholesky factor
of the batch covariance, minus the residual drift.  The passes over the frames run on the device (projection on
the matrix cores, column moments, second moments); the d x d algebra between them (d = number of CVs) is host
arithmetic.  Pinned by tests/golden/whitening.npz, made by importing the reference."""

from __future__ import annotations

import math
import numbers
from typing import Any, Mapping, MutableMapping, Tuple

import numpy as np

from ..device import get_engine

__all__ = ["apply_output_transform", "apply_whitening_from_metadata"]


def _coerce_bool_flag(value: Any) -> bool:
    """None / bool / 0-1 integers / 0.0-1.0 floats / boolean-like strings / 0-d arrays; anything ambiguous raises
    (whitening.py:10-77)."""
    if value is None:
        return False
    if isinstance(value, (bool, np.bool_)):
        return bool(value)
    if isinstance(value, numbers.Integral):
        if int(value) in (0, 1):
            return bool(int(value))
        raise ValueError("already_applied flag integer values must be 0 or 1")
    if isinstance(value, numbers.Real):
        f = float(value)
        if not math.isfinite(f):
            raise ValueError("already_applied flag float must be finite")
        if math.isclose(f, 0.0, rel_tol=0.0, abs_tol=1e-12):
            return False
        if math.isclose(f, 1.0, rel_tol=0.0, abs_tol=1e-12):
            return True
        raise ValueError("already_applied flag float values must be 0.0 or 1.0")
    if isinstance(value, np.ndarray):
        if value.ndim == 0:
            return _coerce_bool_flag(value.item())
        raise TypeError("already_applied flag must be a scalar value; arrays are unsupported")
    if isinstance(value, str):
        v = value.strip().lower()
        if v in {"", "0", "false", "no", "off"}:
            return False
        if v in {"1", "true", "yes", "on"}:
            return True
        raise ValueError("already_applied flag string must be boolean-like (true/false)")
    raise TypeError("already_applied flag must be a boolean, numeric, or string scalar")


def apply_output_transform(Y, mean: Any, W: Any, already_applied) -> np.ndarray:
    """(Y - mean) W, re-centred, then whitened against the batch covariance (when there are more frames than
    CVs) and re-centred again; returned unchanged when the flag says the transform was applied before."""
    arr = np.asarray(Y, dtype=np.float64)
    if _coerce_bool_flag(already_applied):
        return arr
    if mean is None or W is None:
        raise ValueError("Whitening metadata is incomplete: both mean and transform are required")
    mu = np.asarray(mean, dtype=np.float64)
    T = np.asarray(W, dtype=np.float64)
    if mu.ndim != 1:
        raise ValueError("output mean must be a 1D array")
    if T.ndim != 2:
        raise ValueError("output transform must be a 2D matrix")
    if mu.shape[0] != T.shape[0]:
        raise ValueError(f"output mean and transform dimension mismatch: {mu.shape[0]} vs {T.shape[0]}")
    if arr.ndim != 2 or arr.shape[1] != mu.shape[0]:
        raise ValueError(f"projection has incompatible shape for whitening: expected (..., {mu.shape[0]}), got {arr.shape}")
    n, d = arr.shape
    m = T.shape[1]
    if n == 0:
        return np.zeros((0, m))
    if d > 64 or m > 64:
        raise NotImplementedError("output whitening supports up to 64 collective variables")
    eng = get_engine()
    xd = eng.to_device(np.ascontiguousarray(arr))
    ones = eng.to_device(np.ones(d))
    Wfull = np.zeros((d, max(d, m)))
    Wfull[:, :m] = T
    u = eng.project(xd, eng.to_device(mu), ones, eng.to_device(Wfull), m)                 # (Y - mean) W
    drift, _, _ = eng.column_moments(u, ddof=0)
    drift_h = drift.to_host()
    if n <= m:
        return u.to_host() - drift_h[None, :]
    mom = eng.lagged_moments(u, 0, drift, assume_finite=True).to_host()                    # sum (u - drift)(u - drift)'
    cov = 0.5 * mom[:m * m].reshape(m, m) / float(n)
    try:
        L = np.linalg.cholesky(0.5 * (cov + cov.T))
    except np.linalg.LinAlgError as exc:
        raise ValueError("whitening transform produced a singular covariance matrix") from exc
    # np.linalg.solve(L', w')' = w L^-1: the columns combine through the inverse of the lower factor
    M = np.linalg.solve(L.T, np.eye(m)).T
    v = eng.project(u, drift, eng.to_device(np.ones(m)), eng.to_device(np.ascontiguousarray(M)), m)
    last, _, _ = eng.column_moments(v, ddof=0)
    return v.to_host() - last.to_host()[None, :]


def apply_whitening_from_metadata(values, metadata) -> Tuple[np.ndarray, bool]:
    """(whitened, applied): the transform named by metadata["output_mean" / "output_transform"], skipped when
    metadata["output_transform_applied"] is set; a mutable metadata mapping is marked as applied."""
    arr = np.asarray(values, dtype=np.float64)
    if metadata is None:
        raise ValueError("Whitening metadata is required to transform outputs")
    if not isinstance(metadata, Mapping):
        raise TypeError("Whitening metadata must be a mapping with DeepTICA output fields")
    mean, transform = metadata.get("output_mean"), metadata.get("output_transform")
    flag = metadata.get("output_transform_applied")
    if mean is None or transform is None:
        raise ValueError("Whitening metadata must include 'output_mean' and 'output_transform'")
    out = apply_output_transform(arr, mean=mean, W=transform, already_applied=flag)
    applied = not _coerce_bool_flag(flag)
    if applied and isinstance(metadata, MutableMapping):
        metadata["output_transform_applied"] = True
    return out, applied
