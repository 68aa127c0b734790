"""Mirror of the TICA eigenvalue estimator in pmarlo.features.deeptica.core.trainer_api
(S/features/deeptica/core/trainer_api.py:632-656), on the device.

The reference gathers y_t = outputs[idx_t], y_tau = outputs[idx_tau] on the host, centres each with its
own mean, forms C0 = y_t'y_t/(n-1) and Ct = y_t'y_tau/(n-1) with two GEMMs, whitens with
eigh(C0)^-1/2 (eigenvalues clipped at 1e-12) and returns the top eigenvalues of sym(C0^-1/2 Ct C0^-1/2).
Here the two products are ONE pass of the fp64 matrix-core covariance kernel with one-sided second
moments (msm_lagged_moments_onesided) over the resident array -- no gathered copies when the pairs are
(t, t + lag) runs -- and the two small eigensolves run in one launch (msm_onesided_tica_eigenvalues)."""

from __future__ import annotations

from typing import Any, Optional

import numpy as np

from ....device import get_engine

__all__ = ["_estimate_top_eigenvalues", "estimate_top_eigenvalues"]

NUMERIC_MIN_POSITIVE = 1e-12   # S/constants.py:29
_MAX_SEG = 16                   # segments per msm_lagged_moments call


def _pair_runs(idx_t: np.ndarray, idx_tau: np.ndarray):
    """(lag, starts, stops) when the pairs are (t, t + lag), lag >= 0, over strictly increasing t: every run of
    consecutive t = a..b becomes the segment [a, b + lag + 1), whose pairs are exactly those.  Else None."""
    d = idx_tau - idx_t
    lag = int(d[0])
    if lag < 0 or np.any(d != lag) or (idx_t.size > 1 and np.any(np.diff(idx_t) <= 0)) or idx_t[0] < 0:
        return None
    brk = np.flatnonzero(np.diff(idx_t) != 1)
    first = idx_t[np.concatenate([[0], brk + 1])]
    last = idx_t[np.concatenate([brk, [idx_t.size - 1]])]
    return lag, first.astype(np.int64), (last + lag + 1).astype(np.int64)


def estimate_top_eigenvalues(outputs: np.ndarray, idx_t: np.ndarray, idx_tau: np.ndarray, n_out: int,
                             *, engine=None) -> Optional[np.ndarray]:
    outputs = np.asarray(outputs)
    idx_t = np.asarray(idx_t).astype(np.int64, copy=False).ravel()
    idx_tau = np.asarray(idx_tau).astype(np.int64, copy=False).ravel()
    if idx_t.size == 0 or idx_tau.size == 0:
        return None
    if idx_t.size != idx_tau.size:
        raise ValueError("idx_t and idx_tau must have the same length")
    if outputs.ndim != 2:
        raise ValueError(f"outputs must be 2-D, got shape {outputs.shape}")
    if outputs.dtype not in (np.float32, np.float64):
        outputs = outputs.astype(np.float64)
    eng = engine or get_engine()
    n, F = outputs.shape
    runs = _pair_runs(idx_t, idx_tau) if idx_tau.max() < n else None
    if runs is not None:
        lag, starts, stops = runs
        xd = eng.to_device(np.ascontiguousarray(outputs))
    else:
        # arbitrary pairs: gather on the host as the reference does; the stacked copy [y_t; y_tau] is one
        # segment whose pairs (r, r + n_pairs) are the requested ones
        m = idx_t.size
        xd = eng.to_device(np.ascontiguousarray(np.concatenate([outputs[idx_t], outputs[idx_tau]])))
        lag, starts, stops = m, np.array([0], np.int64), np.array([2 * m], np.int64)
    mean, _, cnt = eng.column_moments(xd, ddof=0)            # any shared shift does: the estimator re-centres
    finite = bool(np.all(cnt.to_host() == xd.shape[0]))
    if not finite:
        raise ValueError("estimate_top_eigenvalues: outputs contain non-finite values")
    total = None
    for lo in range(0, len(starts), _MAX_SEG):
        mom = eng.lagged_moments(xd, lag, mean, starts=starts[lo:lo + _MAX_SEG], stops=stops[lo:lo + _MAX_SEG],
                                 assume_finite=True, one_sided=True)
        if len(starts) <= _MAX_SEG:
            total = mom
        else:                                                  # raw moments about one shift add up
            h = mom.to_host()
            total = h if total is None else total + h
    if isinstance(total, np.ndarray):
        total = eng.to_device(total)
    eig = eng.onesided_tica_eigenvalues(total, F, clip=NUMERIC_MIN_POSITIVE).to_host()
    return eig[: min(int(n_out), eig.size)]


def _estimate_top_eigenvalues(outputs: np.ndarray, idx_t: np.ndarray, idx_tau: np.ndarray, cfg: Any) -> Optional[list[float]]:
    """Same signature and return type as the reference: a list of floats (top cfg.n_out, default 2), None for
    empty index arrays.  The reference swallows every exception into None; device failures propagate here."""
    ev = estimate_top_eigenvalues(outputs, np.asarray(idx_t), np.asarray(idx_tau), int(getattr(cfg, "n_out", 2)))
    return None if ev is None else [float(x) for x in ev]
