"""Lag selection by Chapman-Kolmogorov error with coverage / statistics / diagonal-mass guard rails
(SURVEY.md section 8f rank 2): mirror of select_optimal_lag_ck_its and its helpers
(S/markov_state_model/ck_its_selector.py:24-599).

What runs where.  On the device: the count matrices of every candidate lag and every horizon
multiple in one lag-scan pass over the trajectories, the row-normalised T, its stationary vector and
leading eigenvalues, the PCCA+ eigenvectors, the matrix powers of the CK prediction, the relabelled
(macrostate) trajectories and their counts, the L1 error norms, and the reversible maximum-likelihood
estimate behind the timescales / diagonal mass.  On the host: graph connectivity of the k x k count
matrix (scipy, as the reference does), medians, thresholds and the choice itself.

The reference module needs deeptime to import (parity unpinned: checked against the numpy restatement
in oracle/npport.py).  Two behaviours are kept on purpose because results must match the reference:

* PCCA+ is attempted on T = rownorm(C) of the raw counts; deeptime rejects a matrix without detailed
  balance, so that path is only taken for (numerically) reversible counts and the microstate CK test
  is the usual route (:329-391);
* the predicted macrostate kinetics right-multiply by the inverse population matrix,
  (chi' D T^k chi)(chi' D chi + eps I)^-1, as written at :158-189.

Divergences: unassigned frames (label -1) are skipped when trajectories are mapped to macrostates (the
reference's fancy indexing wraps -1 to the last microstate); the eigenvalue-gap rule sees the leading
Ritz values of T rather than its full spectrum; `timescales` holds the ten slowest, not all k - 1."""

from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ..device import get_engine
from .pcca import pcca_like_macrostates

__all__ = ["LagEvaluationResult", "select_optimal_lag_ck_its"]

logger = logging.getLogger("pmarlo")
_EPS = 1e-12     # constants.NUMERIC_MIN_POSITIVE


@dataclass
class LagEvaluationResult:
    lag: int
    ck_error: float
    coverage_fraction: float
    median_count: int
    n_macrostates: int
    n_microstates: int
    passed_sanity: bool
    failure_reason: Optional[str] = None
    timescales: Optional[np.ndarray] = None
    eigenvalue_gap: Optional[float] = None
    diag_mass: Optional[float] = None


class _Trajectories:
    """The discrete trajectories on the device, one segment per trajectory (a pair is counted when both
    of its frames carry a valid label, whatever lies between them: _count_transitions :70-84)."""

    def __init__(self, dtrajs: Sequence[np.ndarray], n_states: int):
        self.eng = get_engine()
        self.k = int(n_states)
        arrs = [np.asarray(d).astype(np.int32, copy=False).ravel() for d in dtrajs]
        bounds = np.cumsum([0] + [a.size for a in arrs])
        self.starts = bounds[:-1].astype(np.int64)
        self.stops = bounds[1:].astype(np.int64)
        self.labels = self.eng.to_device(np.concatenate(arrs))

    def counts(self, lags: Sequence[int], labels=None, k: int | None = None):
        """int64 [len(lags), k, k] on the device."""
        out, _ = self.eng.count_transitions_lagscan(self.labels if labels is None else labels, self.k if k is None else k,
                                                    [int(v) for v in lags], starts=self.starts, stops=self.stops)
        return out


def _coverage_fraction(C: np.ndarray) -> float:
    """Share of the states inside the largest connected set of the undirected count graph (:86-102)."""
    from scipy.sparse.csgraph import connected_components

    if C.size == 0:
        return 0.0
    n_comp, lab = connected_components(((C + C.T) > 0).astype(int), directed=False, return_labels=True)
    if n_comp == 0:
        return 0.0
    return float(np.bincount(lab).max()) / float(C.shape[0])


def _median_count(C: np.ndarray) -> int:
    tot = C.sum(axis=0) + C.sum(axis=1)
    seen = tot[tot > 0]
    return int(np.median(seen)) if seen.size else 0


def _auto_macrostates(evals_desc: np.ndarray, lo: int, hi: int) -> int:
    """Largest gap lambda_{m-1} - lambda_m for m in [lo, hi] (:117-155); evals sorted descending."""
    if evals_desc.size < lo + 1:
        return lo
    best, width = lo, 0.0
    for m in range(lo, min(hi + 1, evals_desc.size)):
        gap = float(evals_desc[m - 1] - evals_desc[m])
        if gap > width:
            best, width = m, gap
    return best


def _row_normalize(eng, counts):
    """deeptime's transition_matrix_non_reversible behind _row_normalize (_msm_utils.py:70-75): C / rowsum,
    ValueError when a row is empty (the caller's try/except turns that into a failed evaluation)."""
    out = eng.transition_matrix(counts, mode=0)
    low = float(np.min(out["rowsum"].to_host()))
    if low <= 0:
        raise ValueError(f"Transition matrix has row sum of {low}. Must have strictly positive row sums.")
    return out["T"]


def _l1_relative(eng, pred, obs) -> float:
    l1, ref, _ = eng.diff_norms(pred, obs)
    return float("inf") if ref < _EPS else float(l1 / ref)


def _largest_strong_set(C: np.ndarray) -> np.ndarray:
    from scipy.sparse.csgraph import connected_components

    _, lab = connected_components((C > 0).astype(int), directed=True, connection="strong", return_labels=True)
    sizes = np.bincount(lab)
    return np.flatnonzero(lab == int(np.argmax(sizes)))


def _reversible_summary(eng, C: np.ndarray, lag: int, n_ts: int = 10):
    """Timescales and mean diagonal of deeptime's MaximumLikelihoodMSM(reversible=True) on the largest
    connected set (:393-401): the engine's reversible estimator on the same set."""
    act = _largest_strong_set(C)
    sub = np.ascontiguousarray(C[np.ix_(act, act)], dtype=np.float64)
    if act.size == 0 or sub.sum() <= 0:
        raise ValueError("no connected transitions")
    if act.size == 1:
        return np.empty((0,), dtype=float), 1.0
    out = eng.reversible_mle(eng.to_device(sub))
    n = act.size
    want = int(min(n_ts, n - 1))
    spec = eng.spectrum(out["T"], n_its=0, n_watch=want + 1, want_pi=False, allow_unconverged=True)
    # deeptime's timescales(): -lag / ln|lambda_i| of the non-unit eigenvalues in descending magnitude
    mag = np.abs(spec["ritz"][0][1:1 + want])
    with np.errstate(divide="ignore", invalid="ignore"):
        ts = -float(lag) / np.log(np.clip(mag, _EPS, 1.0 - _EPS))
    diag = float(np.trace(out["T"].to_host()) / n)
    return np.asarray(ts, dtype=float), diag


def _evaluate_lag(tr: _Trajectories, lag: int, horizons: List[int], C_by_lag: dict, coverage_threshold: float,
                  min_median_count: int, diag_mass_threshold: float) -> LagEvaluationResult:
    eng, k = tr.eng, tr.k
    try:
        C_dev = C_by_lag[lag]
        C = C_dev.to_host().astype(float)
        coverage, median = _coverage_fraction(C), _median_count(C)
        if coverage < coverage_threshold:
            why = f"Coverage {coverage:.2%} < {coverage_threshold:.2%}"
        elif median < min_median_count:
            why = f"Median count {median} < {min_median_count}"
        else:
            why = None
        if why is not None:
            return LagEvaluationResult(lag, float("inf"), coverage, median, 0, k, False, why)
        T = _row_normalize(eng, C_dev)
        spec = eng.spectrum(T, n_its=0, p=min(k, 16), n_watch=min(k, 7), allow_unconverged=True)
        pi = spec["pi"].view((k,))
        evals = np.sort(np.real(spec["ritz"][0][:min(k, spec["p"])]))[::-1]
        n_cand = _auto_macrostates(evals, 2, 6) if k >= 2 else 2
        T_host = T.to_host()
        macro = pcca_like_macrostates(T_host, n_macrostates=n_cand)
        n_macro, gap = 0, None
        if macro is not None:
            n_macro = n_cand
            pi_h = pi.to_host()
            chi = np.zeros((k, n_macro))
            chi[np.arange(k), macro] = 1.0
            left = eng.to_device(np.ascontiguousarray(chi.T * pi_h[None, :]))        # chi' diag(pi)   [m, k]
            chi_d = eng.to_device(chi)
            pops = (chi.T * pi_h[None, :]) @ chi + np.eye(n_macro) * _EPS
            inv_pops = np.linalg.inv(pops)
            macro_traj = eng.relabel(tr.labels, macro.astype(np.int32))
            obs = tr.counts([lag * h for h in horizons], labels=macro_traj, k=n_macro)
            worst, power, lt = 0.0, 0, left
            for h_i, h in enumerate(horizons):
                while power < h:                                                         # (chi' D) T^h, m x k
                    lt = eng.gemm(lt, T)
                    power += 1
                num = eng.gemm(lt, chi_d).to_host()
                T_pred = eng.to_device(np.ascontiguousarray(num @ inv_pops))
                T_obs = _row_normalize(eng, obs.view((n_macro, n_macro), offset_elems=h_i * n_macro * n_macro))
                worst = max(worst, _l1_relative(eng, T_pred, T_obs))
            ck = worst
            if evals.size > n_macro:
                gap = float(evals[n_macro - 1] - evals[n_macro])
        else:
            logger.warning("[CK-ITS] PCCA+ failed for lag %d, using microstate CK test fallback", lag)
            worst, power, Tp = 0.0, 1, T
            for h in sorted(set(horizons)):
                while power < h:
                    Tp = eng.gemm(Tp, T)
                    power += 1
                T_obs = _row_normalize(eng, C_by_lag[lag * h])
                worst = max(worst, _l1_relative(eng, Tp, T_obs))
            ck = worst
        timescales, diag = None, float("nan")
        try:
            timescales, diag = _reversible_summary(eng, C, lag)
        except Exception as exc:  # the reference logs and carries on without timescales (:402-406)
            logger.warning("[CK-ITS] Failed to compute timescales for lag %d: %s", lag, exc)
        why = None
        if not (np.isfinite(diag) and diag >= diag_mass_threshold):
            why = (f"Diagonal mass {diag:.3f} < threshold {diag_mass_threshold:.3f}" if np.isfinite(diag)
                   else "Diagonal mass undefined")
        return LagEvaluationResult(lag, ck, coverage, median, n_macro, k, why is None, why, timescales, gap, diag)
    except Exception as exc:  # one bad lag must not stop the scan (:450-460)
        logger.error("[CK-ITS] Failed to evaluate lag %d: %s", lag, exc, exc_info=True)
        return LagEvaluationResult(lag, float("inf"), 0.0, 0, 0, k, False, f"Exception: {exc}")


def select_optimal_lag_ck_its(dtrajs: Sequence[np.ndarray], tau_candidates: Optional[List[int]] = None,
                              horizons: Optional[List[int]] = None, ck_threshold: float = 0.15,
                              coverage_threshold: float = 0.98, min_median_count: int = 100,
                              diag_mass_threshold: float = 0.6) -> Tuple[int, List[LagEvaluationResult]]:
    """Smallest candidate lag whose CK error is <= ck_threshold among those passing coverage, median-count
    and diagonal-mass checks; else the passing lag with the smallest error; else the smallest candidate."""
    if not dtrajs or len(dtrajs) == 0:
        raise ValueError("No discrete trajectories provided")
    usable = [np.asarray(t) for t in dtrajs if t is not None and np.asarray(t).size > 0]
    if not usable:
        raise ValueError("Discrete trajectories contain no frames for CK analysis; "
                         "provide trajectories with at least two time steps.")
    tau_candidates = [25, 50, 75, 100] if tau_candidates is None else list(tau_candidates)
    horizons = [1, 2, 3, 4, 5] if horizons is None else [int(h) for h in horizons]
    longest = max(int(t.size) for t in usable) - 1
    valid = [int(t) for t in tau_candidates if t <= longest]
    ignored = [int(t) for t in tau_candidates if t > longest]
    if ignored:
        logger.warning("[CK-ITS] Ignoring %d tau candidates that exceed available length (max supported lag=%d): %s",
                       len(ignored), longest, ignored)
    if not valid:
        raise ValueError("All tau candidates exceed the available trajectory length "
                         f"(max supported lag {longest}). Provide smaller lag values or shorter horizons.")
    n_states = int(max(np.max(t) for t in usable)) + 1
    tr = _Trajectories(usable, n_states)
    every = sorted({lag * h for lag in valid for h in set(horizons) | {1}})
    stack = tr.counts(every)                                   # one pass over the frames for all of them
    C_by_lag = {lv: stack.view((n_states, n_states), offset_elems=i * n_states * n_states) for i, lv in enumerate(every)}
    evaluations = [_evaluate_lag(tr, lag, horizons, C_by_lag, coverage_threshold, min_median_count, diag_mass_threshold)
                   for lag in sorted(valid)]
    for res in sorted(evaluations, key=lambda r: r.lag):
        if res.passed_sanity and res.ck_error <= ck_threshold:
            return res.lag, evaluations
    passing = [r for r in evaluations if r.passed_sanity]
    if passing:
        return min(passing, key=lambda r: r.ck_error).lag, evaluations
    return min(tau_candidates), evaluations
