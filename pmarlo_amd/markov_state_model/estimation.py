"""MSM estimation hooks: mirror of EstimationMixin (S/markov_state_model/_estimation.py:
116-156 _count_transitions_deeptime, :158-188 _finalize_transition_and_stationary,
:211-220 _compute_free_energies) and ensure_connected_counts (S/utils/msm_utils.py:129-167)."""

from __future__ import annotations

from typing import Sequence

import numpy as np

from ..device import get_engine
from .results import ConnectedCountResult, MSMEstimate

__all__ = ["count_transitions", "ensure_connected_counts", "finalize_transition_and_stationary", "build_msm",
           "compute_free_energies"]

NUMERIC_MIN_POSITIVE = 1e-12
NUMERIC_DIRICHLET_ALPHA = 1e-3


def _concat_dtrajs(dtrajs: Sequence[np.ndarray], n_states: int):
    """Concatenate label arrays; out-of-range labels split segments exactly like the reference's
    Python loop (:133-145) because the count kernel skips any pair touching an invalid label...
    a pair may not BRIDGE an invalid frame though, so invalid frames also cut segments."""
    arrays, segs, off = [], [], 0
    for d in dtrajs:
        a = np.asarray(d)
        if a.size == 0:
            continue
        a = a.astype(np.int32, copy=False)
        valid = (a >= 0) & (a < n_states)
        if valid.all():
            segs.append((off, off + a.size))
        else:
            edges = np.flatnonzero(np.diff(np.concatenate([[0], valid.view(np.int8), [0]])))
            for s, e in zip(edges[::2], edges[1::2]):
                segs.append((off + int(s), off + int(e)))
        arrays.append(a)
        off += a.size
    labels = np.concatenate(arrays) if arrays else np.zeros(0, np.int32)
    return labels, segs


def count_transitions(dtrajs: Sequence[np.ndarray], n_states: int, *, lag: int, count_mode: str = "sliding") -> np.ndarray:
    """(n, n) float64 sliding-window counts; ``"strided"`` is mapped to sliding as the reference does (:152)."""
    if count_mode not in ("sliding", "strided"):
        raise ValueError(f"unsupported count_mode {count_mode!r} on the accelerated path")
    labels, segs = _concat_dtrajs(dtrajs, n_states)
    if labels.size == 0 or not segs:
        return np.zeros((n_states, n_states), dtype=float)
    eng = get_engine()
    starts = np.asarray([s for s, _ in segs], np.int64)
    stops = np.asarray([e for _, e in segs], np.int64)
    counts, _ = eng.count_transitions(eng.to_device(labels), n_states, int(max(1, lag)), starts=starts, stops=stops)
    return counts.to_host().astype(float)


def ensure_connected_counts(C: np.ndarray, alpha: float = NUMERIC_DIRICHLET_ALPHA,
                            epsilon: float = NUMERIC_MIN_POSITIVE) -> ConnectedCountResult:
    C = np.asarray(C)
    if C.ndim != 2 or C.shape[0] != C.shape[1]:
        raise ValueError("count matrix must be square")
    totals = C.sum(axis=1) + C.sum(axis=0)
    active = np.where(totals > epsilon)[0]
    if active.size == 0:
        return ConnectedCountResult(np.empty((0, 0), dtype=float), active)
    return ConnectedCountResult(C[np.ix_(active, active)].astype(float) + float(alpha), active)


def finalize_transition_and_stationary(counts: np.ndarray) -> MSMEstimate:
    """Regularised active set, row-normalised T (ML-MSM, reversible=False), stationary vector;
    T_full = I outside the active block, pi_full = 0 there.  All numerics on the device."""
    C = np.ascontiguousarray(counts, dtype=np.float64)
    n = C.shape[0]
    eng = get_engine()
    out = eng.transition_matrix(eng.to_device(C), mode=1)
    ka = int(out["n_active"].to_host()[0])
    cm = np.zeros((n, n))
    if ka == 0:
        return MSMEstimate(cm, np.eye(n), np.zeros(n), np.zeros(0, dtype=int))
    spec = eng.spectrum(out["T"], n=out["n_active"], n_its=0)
    T_full, pi_full = eng.embed_full(out["T"], out["inv_map"], spec["pi"])
    active = out["active"].to_host()[:ka].astype(int)
    cm[np.ix_(active, active)] = C[np.ix_(active, active)] + NUMERIC_DIRICHLET_ALPHA
    return MSMEstimate(cm, T_full.to_host(), pi_full.to_host(), active)


def compute_free_energies(stationary_distribution: np.ndarray, temperature: float = 300.0) -> np.ndarray:
    kT = 1.380649e-23 * temperature * 6.02214076e23 / 1000.0  # kJ/mol (scipy.constants values)
    F = -kT * np.log(np.maximum(stationary_distribution, NUMERIC_MIN_POSITIVE))
    return F - np.min(F)


def build_msm(dtrajs: Sequence[np.ndarray], n_states: int, lag_time: int = 20, count_mode: str = "sliding",
              temperature: float = 300.0) -> MSMEstimate:
    """build_msm (:50-100) minus the orchestration: counts -> T, pi -> free energies."""
    lens = [len(d) for d in dtrajs if len(d)]
    lag = int(max(1, lag_time))
    if lens and lag > min(lens) - 1 > 0:
        lag = min(lens) - 1  # _validate_and_cap_lag (:102-114)
    est = finalize_transition_and_stationary(count_transitions(dtrajs, n_states, lag=lag, count_mode=count_mode))
    est.free_energies = compute_free_energies(est.stationary_distribution, temperature)
    return est


def fit_reversible_msm(counts: np.ndarray, *, maxerr: float = 1e-8, maxiter: int = 1_000_000):
    """ensure_connected_counts + MaximumLikelihoodMSM(reversible=True) + _expand_results
    (_fit_msm_deeptime / _expand_results, S/markov_state_model/_msm_utils.py:210-281): T = I and
    pi = 0 outside the active block.  The fixed-point iteration runs on the device."""
    C = np.asarray(counts, dtype=np.float64)
    n = C.shape[0]
    res = ensure_connected_counts(C)
    T_full, pi_full = np.eye(n), np.zeros(n)
    if res.counts.size == 0:
        return T_full, pi_full, res.active
    eng = get_engine()
    out = eng.reversible_mle(eng.to_device(np.ascontiguousarray(res.counts)), maxerr=maxerr, maxiter=maxiter)
    T_full[np.ix_(res.active, res.active)] = out["T"].to_host()
    pi_full[res.active] = out["pi"].to_host()
    return T_full, pi_full, res.active


def build_simple_msm(dtrajs: Sequence[np.ndarray], n_states: int | None = None, lag: int = 20,
                     count_mode: str = "sliding") -> tuple[np.ndarray, np.ndarray]:
    """(transition_matrix, stationary_distribution) of the reversible estimate
    (build_simple_msm, S/markov_state_model/_msm_utils.py:163-187)."""
    if not dtrajs:
        return np.empty((0, 0), dtype=float), np.empty((0,), dtype=float)
    if n_states is None:                                 # _infer_n_states :190-207: negative labels = unassigned
        top = max((int(np.max(d)) for d in dtrajs if len(d)), default=-1)
        n_states = top + 1 if top >= 0 else 0
    n_states = int(n_states)
    if n_states == 0:
        return np.empty((0, 0), dtype=float), np.empty((0,), dtype=float)
    C = count_transitions(dtrajs, n_states, lag=int(max(1, lag)), count_mode=count_mode)
    T, pi, _ = fit_reversible_msm(C)
    check_transition_matrix(T, pi)
    return T, pi


def check_transition_matrix(T: np.ndarray, pi: np.ndarray, *, row_tol: float = 1e-12, stat_tol: float = 1e-8) -> None:
    """Validate a transition matrix with its stationary vector (S/utils/msm_utils.py:168-299): non-negative
    entries, unit row sums, pi T = pi, and agreement of pi with the stationary vector of T itself (from the device
    solver) -- except on states the chain cannot reach from the support of pi, which mark T as reducible instead
    of failing.  Raises ValueError; returns nothing."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components

    T = np.asarray(T, dtype=float)
    pi = np.asarray(pi, dtype=float)
    if T.ndim != 2 or T.shape[0] != T.shape[1]:
        raise ValueError("transition matrix must be square")
    if pi.shape != (T.shape[0],):
        raise ValueError("stationary distribution size mismatch")
    if T.size == 0:
        return
    if np.any(T < 0.0):
        raise ValueError("Negative probabilities in transition matrix")
    if np.abs(T.sum(axis=1) - 1.0).max() >= T.shape[0] * row_tol:       # deeptime's is_transition_matrix(T, tol)
        raise ValueError("transition matrix fails stochasticity checks")
    total = float(pi.sum())
    if not np.isfinite(total) or total <= 0:
        raise ValueError("stationary distribution must be normalisable")
    p = pi / total
    residual = float(np.max(np.abs(p @ T - p)))
    if residual > stat_tol:
        raise ValueError(f"provided stationary distribution fails invariance check (max residual {residual})")
    rate = 1e-6                                                         # constants.NUMERIC_MIN_RATE
    reducible = False
    if T.shape[0] > 1:
        n_comp, lab = connected_components(csr_matrix((T > rate).astype(int)), directed=True, connection="strong")
        if n_comp > 1:
            closed = [not np.any((T[lab == c] > rate)[:, lab != c]) for c in range(n_comp)]
            reducible = sum(closed) > 1
    eng = get_engine()
    ref = eng.spectrum(eng.to_device(np.ascontiguousarray(T)), n_its=0, allow_unconverged=True)["pi"].to_host().ravel()
    diff = np.abs(p - ref)
    support = p > stat_tol
    if support.any():
        unreachable = ~support & ~np.any(T[support] > rate, axis=0)
    else:
        unreachable = np.ones(diff.shape, dtype=bool)
    if unreachable.any():
        diff[unreachable] = 0.0
        reducible = True
    worst = float(diff.max()) if diff.size else 0.0
    if worst > stat_tol and not reducible:
        raise ValueError(f"Stationary distribution mismatch at state {int(np.argmax(diff))} with error {worst}")
