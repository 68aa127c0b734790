"""Implied timescales: mirror of ITSMixin (S/markov_state_model/_its.py:137-192, 272-357,
543-668) and utils.safe_timescales (S/markov_state_model/utils.py:17-57).

The reference's ITS is the median (and a percentile band) over n_samples Bayesian posterior
matrices drawn by deeptime's C++ sampler, whose random stream cannot be reproduced outside
deeptime (SURVEY.md hard part 4).  This engine keeps the definition and replaces the sampler:

* n_samples <= 1: _summarize_its_stats applied to the single maximum-likelihood matrix
  T(tau) = rownorm(C(tau) + alpha) on the active set (deterministic; zero-width intervals);
* n_samples > 1: the same statistics over n_samples draws of the closed-form posterior of that
  estimator (independent Dirichlet rows, msm_sample_transition_matrices), all lags and samples
  in one batched spectrum solve; median and percentile band as _summarize_its_stats :606-625."""

from __future__ import annotations

from typing import Sequence

import logging

import numpy as np

from ..device import get_engine
from .estimation import _concat_dtrajs
from .results import ITSResult

logger = logging.getLogger("pmarlo")

__all__ = ["format_lag_window_ps", "safe_timescales", "compute_implied_timescales", "deterministic_its_from_counts", "detect_timescale_plateau",
           "select_lag_from_its",
           "candidate_lag_ladder", "DEFAULT_ITS_LAGS"]

EPS = 1e-12


def format_lag_window_ps(window: tuple[float, float]) -> str:
    """"start\u2013end ps" with three decimals (S/markov_state_model/utils.py:60-64)."""
    return f"{window[0]:.3f}\u2013{window[1]:.3f} ps"


def safe_timescales(lag: float, eigvals, eps: float = EPS) -> np.ndarray:
    eig = np.asarray(eigvals)
    if eig.size == 0:
        return np.empty_like(eig, dtype=np.float64)
    ec = eig.astype(np.complex128, copy=False)
    mag = np.abs(ec)
    with np.errstate(divide="ignore", invalid="ignore"):
        ts = np.asarray(-float(lag) / np.log(np.clip(mag, eps, 1 - eps)), dtype=np.float64)
    invalid = ~np.isfinite(mag) | (mag <= 0) | (mag >= 1)
    invalid |= np.isclose(ec.imag, 0.0) & ((ec.real <= 0.0) | (ec.real >= 1.0))
    ts[invalid] = np.nan
    return ts


# _its_default_lag_times, S/markov_state_model/_its.py:421-447
DEFAULT_ITS_LAGS = (1, 2, 3, 5, 8, 10, 15, 20, 30, 40, 50, 75, 80, 100, 150, 160, 200, 320, 640, 1280)
# curated ladder of utils/msm_utils.candidate_lag_ladder (S/utils/msm_utils.py:54-81)
_LADDER = (1, 2, 3, 5, 8, 10, 15, 20, 30, 40, 50, 75, 80, 100, 150, 160, 200, 300, 320, 500, 640, 750, 1000, 1280,
           1500, 2000)
_BATCH_BYTES = 8 << 30      # sampled matrices held at once (a lag scan larger than this runs in slices of lags)


def candidate_lag_ladder(min_lag: int = 1, max_lag: int = 200, n_candidates: int | None = None) -> list[int]:
    """Curated lag ladder within [min_lag, max_lag], optionally thinned to n_candidates points spread
    evenly over the ladder with both ends kept (S/utils/msm_utils.py:21-105)."""
    lo, hi = int(min_lag), int(max_lag)
    if lo < 1:
        raise ValueError("min_lag must be >= 1")
    if hi < lo:
        raise ValueError("max_lag must be >= min_lag")
    if n_candidates is not None and n_candidates < 1:
        raise ValueError("n_candidates must be positive")
    inside = [v for v in _LADDER if lo <= v <= hi]
    if not inside:
        raise ValueError(f"No predefined lag values available in range [{lo}, {hi}]")
    if n_candidates is None or n_candidates >= len(inside):
        return inside
    if n_candidates <= 2:
        return [inside[0]] if n_candidates == 1 else [inside[0], inside[-1]]
    spacing = (len(inside) - 1) / (n_candidates - 1)
    idx = sorted({int(round(q * spacing)) for q in range(n_candidates)})
    idx[0], idx[-1] = 0, len(inside) - 1
    return [inside[q] for q in idx]


def detect_timescale_plateau(lag_times, timescales, m: int, epsilon: float) -> tuple[float, float] | None:
    """Longest run of >= m consecutive lags over which the slowest timescale varies by at most
    epsilon times its mean; (first lag, last lag) of the earliest such run, or None
    (ITSMixin._detect_timescale_plateau, S/markov_state_model/_its.py:803-839)."""
    ts = np.asarray(timescales, dtype=float)
    lags = np.asarray(lag_times, dtype=float)
    if ts.size == 0 or lags.size == 0:
        return None
    slow = ts[:, 0] if ts.ndim > 1 else ts
    need = max(1, int(m))
    best, span = 0, None
    for a in range(slow.shape[0]):
        lo = hi = tot = None
        for b in range(a, slow.shape[0]):
            v = float(slow[b])
            if not np.isfinite(v):
                break                                   # a gap ends every window that starts at a
            lo, hi, tot = (v, v, v) if lo is None else (min(lo, v), max(hi, v), tot + v)
            width = b - a + 1
            if width < need:
                continue
            mean = tot / width
            if mean > 0 and np.isfinite(mean) and hi - lo <= float(epsilon) * mean and width > best:
                best, span = width, (float(lags[a]), float(lags[b]))
    return span


def select_lag_from_its(lag_times, timescales, *, min_lag_idx: int = 3, plateau_threshold: float = 0.15) -> int:
    """First lag (from index min_lag_idx) at which the slowest timescale changes by less than
    plateau_threshold relative to the previous lag, confirmed by the next step staying below 1.5x the
    threshold; otherwise the lag of the largest timescale in the second half of the scan
    (select_lag_from_its, S/markov_state_model/_msm_utils.py:302-399).  Empty input: 10."""
    lags = np.asarray(lag_times)
    ts = np.asarray(timescales, dtype=float)
    if lags.size == 0 or ts.size == 0:
        return 10
    slow = ts[:, 0] if ts.ndim > 1 else ts
    ok = np.isfinite(slow) & (slow > 0)
    if not ok.any():
        return 10
    n = slow.shape[0]
    first = max(1, int(min_lag_idx))
    if first >= n:
        first = max(1, n // 4)
    for q in range(first, n):
        if not (ok[q] and ok[q - 1]):
            continue
        if abs((slow[q] - slow[q - 1]) / slow[q - 1]) >= plateau_threshold:
            continue
        if q + 1 < n and ok[q + 1]:
            if abs((slow[q + 1] - slow[q]) / slow[q]) < 1.5 * plateau_threshold:
                return int(lags[q])
        else:
            return int(lags[q])
    half = n // 2
    if ok[half:].any():
        return int(lags[half + int(np.argmax(np.where(ok[half:], slow[half:], -np.inf)))])
    return int(lags[n // 2])


def deterministic_its_from_counts(counts: np.ndarray, lag: int, n_timescales: int, *, reference_quirk: bool = False):
    """(eigenvalues, timescales, rates) of the symmetrised estimate T = rownorm((C + C') / 2): the fall-back of
    ITSMixin._deterministic_its_from_counts (S/markov_state_model/_its.py:742-801).

    reference_quirk=False (default) is its mathematically intended form: the spectrum of T (real, T being
    reversible with pi ~ row sums of C + C'), on the device.
    reference_quirk=True reproduces what the reference returns.  It takes "pi" from the row sums of T
    (:753-754), which are identically 1, so deeptime's similarity transform diag(pi)^1/2 T diag(pi)^-1/2 is T
    itself and its dense reversible branch calls numpy.linalg.eigvalsh on a matrix that is NOT symmetric --
    LAPACK then reads the lower triangle only.  The quirk's values are therefore the eigenvalues of the symmetric
    matrix built from the lower triangle of T, ordered by decreasing magnitude (deeptime), the reported
    eigenvalues re-sorted by decreasing real part (:763) while the timescales keep deeptime's order (:778-787).
    Solved by msm_eigh (k <= 256)."""
    C = np.asarray(counts, dtype=np.float64)
    n = int(n_timescales)
    ev, ts = np.zeros(n), np.full(n, np.nan)
    k = C.shape[0]
    if n > 0 and k > 0 and reference_quirk:
        if k > 256:
            raise NotImplementedError("reference_quirk=True solves a dense k x k symmetric problem on the device: k <= 256")
        eng = get_engine()
        Crev = 0.5 * (C + C.T)
        row = Crev.sum(axis=1, keepdims=True)
        T = Crev / np.where(row == 0, 1.0, row)
        low = np.tril(T)
        w = eng.eigh(eng.to_device(np.ascontiguousarray(low + np.tril(T, -1).T)), want_vectors=False)[0].to_host()
        w = w[np.argsort(np.abs(w), kind="stable")[::-1]]            # deeptime: decreasing magnitude
        k_eval = None if n + 1 > k else n + 1
        wk = w if k_eval is None else w[:k_eval]
        slow = np.sort(wk)[::-1][1:1 + n]                              # :763-768: re-sorted by decreasing real part
        ev[:slow.shape[0]] = np.clip(np.abs(slow), 1e-12, 1.0 - 1e-12)
        k_times = None if k_eval is None else min(k, n + 1)
        wt = w if k_times is None else w[:k_times]
        tsr = np.zeros(wt.shape[0])
        one = np.isclose(np.abs(wt), 1.0, rtol=0.0, atol=1e-14)        # deeptime timescales_from_eigenvalues
        tsr[one] = np.inf
        with np.errstate(divide="ignore"):
            tsr[~one] = -float(max(1, int(lag))) / np.log(np.abs(wt[~one]))
        cut = tsr[1:1 + n]
        ts[:cut.shape[0]] = cut
    elif n > 0 and k > 0:
        eng = get_engine()
        T = eng.transition_matrix(eng.to_device(np.ascontiguousarray(0.5 * (C + C.T))), mode=0)["T"]
        want = min(n, max(k - 1, 0))
        if want > 0:
            spec = eng.spectrum(T, n_its=want, lags=[float(max(1, int(lag)))], want_pi=False, allow_unconverged=True)
            ev[:want] = np.nan_to_num(spec["its_eig"][0], nan=0.0)
            ts[:want] = spec["its_ts"][0]
    with np.errstate(divide="ignore", invalid="ignore"):
        rates = np.where(np.isfinite(ts), 1.0 / ts, np.nan)
    return ev, ts, rates


def _posterior_summary(ev: np.ndarray, ts: np.ndarray, q_low: float, q_high: float):
    """Median and percentile band over the sample axis (axis 1) of [L, S, n] arrays
    (_summarize_its_stats, S/markov_state_model/_its.py:606-625)."""
    import warnings

    with np.errstate(divide="ignore", invalid="ignore"):
        rate = np.where(np.isfinite(ts), 1.0 / ts, np.nan)
    out = []
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", category=RuntimeWarning)
        for arr in (ev, ts, rate):
            out.append(np.nanmedian(arr, axis=1))
            out.append(np.stack([np.nanpercentile(arr, q_low, axis=1), np.nanpercentile(arr, q_high, axis=1)], axis=-1))
    return out


def compute_implied_timescales(dtrajs: Sequence[np.ndarray], n_states: int, lag_times: Sequence[int] | None = None,
                               n_timescales: int = 5, *, n_samples: int = 100, ci: float = 0.95,
                               dirichlet_alpha: float = 1e-3, plateau_m: int | None = None,
                               plateau_epsilon: float = 0.1, time_per_frame_ps: float | None = None,
                               random_state: int | None = None, return_samples: bool = False,
                               effective_frames: int | None = None) -> ITSResult:
    """Lag scan on the device (ITSMixin.compute_implied_timescales, S/markov_state_model/_its.py:137-192):
    batched counts for all lags, one packed transition matrix per lag, n_samples posterior matrices
    per lag, one batched spectrum solve over lags x samples, median / percentile band on the host.

    Input rules of _validate_its_inputs (:453-524): no trajectories, or a shortest trajectory with fewer than two
    frames (an EMPTY trajectory counts: it makes max_valid_lag = -1), give the empty result; lags above
    min(len) - 1 are dropped; a largest lag >= `effective_frames` (the mixin's attribute of that name) raises.
    The confidence bands come from independent Dirichlet rows on C_active + alpha (the closed-form posterior of the
    non-reversible estimator this engine fits), not from deeptime's reversible BayesianMSM sampler: medians and
    bands differ from the reference's systematically where detailed balance matters."""
    n = int(n_timescales)
    empty = ITSResult(lag_times=np.array([], dtype=int), eigenvalues=np.empty((0, n)),
                      eigenvalues_ci=np.empty((0, n, 2)), timescales=np.empty((0, n)),
                      timescales_ci=np.empty((0, n, 2)), rates=np.empty((0, n)), rates_ci=np.empty((0, n, 2)))
    if dtrajs is None or len(dtrajs) == 0:
        logger.warning("No trajectories available for implied timescales")
        return empty
    max_valid = min(len(d) for d in dtrajs) - 1           # :476 -- an empty trajectory is not skipped
    if max_valid < 1:
        logger.warning("Trajectories too short for implied timescales")
        return empty
    wanted = DEFAULT_ITS_LAGS if lag_times is None else [int(max(1, v)) for v in lag_times]
    if any(int(v) > max_valid for v in wanted):
        logger.warning("Capping lag times above max_valid_lag=%s", max_valid)
    lags = [int(v) for v in wanted if 1 <= int(v) <= max_valid]  # :493-500
    if not lags:
        logger.warning("No valid lag times after capping")
        return empty
    if effective_frames is not None and effective_frames > 0 and max(lags) >= effective_frames:   # :517-522
        raise ValueError(f"Maximum lag {max(lags)} exceeds available effective frames {effective_frames}")
    labels, segs = _concat_dtrajs(dtrajs, n_states)
    eng = get_engine()
    k, L = int(n_states), len(lags)
    S = int(n_samples) if int(n_samples) > 1 else 0
    starts = np.asarray([s for s, _ in segs], np.int64)
    stops = np.asarray([e for _, e in segs], np.int64)
    counts, _ = eng.count_transitions_lagscan(eng.to_device(labels), k, lags, starts=starts, stops=stops)
    Tb = eng.empty((L, k, k), np.float64)
    nb = eng.empty((L,), np.int32)
    act = eng.empty((L, k), np.int32)
    from .._lib import check, lib

    rows, inv = eng.empty((k,), np.float64), eng.empty((k,), np.int32)
    for i in range(L):  # per-lag active set + alpha, packed in place (device to device)
        check(lib.msm_transition_matrix(eng.handle, counts.ptr + i * k * k * 8, 0, k, 1, float(dirichlet_alpha), 1e-12,
                                        Tb.ptr + i * k * k * 8, act.ptr + i * k * 4, inv.ptr, nb.ptr + 4 * i, rows.ptr,
                                        None), eng.handle)
    if not S:
        spec = eng.spectrum(Tb, n=nb, n_its=n, lags=[float(v) for v in lags], want_pi=False)
        ev, ts = spec["its_eig"], spec["its_ts"]
        with np.errstate(divide="ignore", invalid="ignore"):
            rates = np.where(np.isfinite(ts), 1.0 / ts, np.nan)
        band = lambda a: np.stack([a, a], axis=-1)  # noqa: E731  deterministic: zero-width intervals
        ev_ci, ts_ci, rate_ci = band(ev), band(ts), band(rates)
        samples = None
    else:
        seed = 0 if random_state is None else int(random_state)
        per = max(1, min(L, _BATCH_BYTES // (S * k * k * 8)))
        ev_s, ts_s = np.empty((L, S, n)), np.empty((L, S, n))
        nb_h = nb.to_host()
        for a in range(0, L, per):
            b = min(L, a + per)
            batch = eng.empty(((b - a) * S, k, k), np.float64)
            for i in range(a, b):
                # every lag has its own stream: sample numbers i*S .. i*S + S - 1
                eng.sample_transition_matrices(
                    counts.view((k, k), offset_elems=i * k * k), act.view((k,), offset_elems=i * k),
                    nb.view((1,), offset_elems=i), alpha=float(dirichlet_alpha), seed=seed, n_samples=S,
                    first_sample=i * S, out=batch.view((S, k, k), offset_elems=(i - a) * S * k * k))
            nrep = eng.to_device(np.repeat(nb_h[a:b], S).astype(np.int32))
            # posterior samples have a noisy bulk right under the watched eigenvalues: the widest subspace from the start
            # (the solver would get there after two launches and a restart)
            spec = eng.spectrum(batch, n=nrep, n_its=n, lags=np.repeat(np.asarray(lags[a:b], np.float64), S),
                                want_pi=False, allow_unconverged=True, p=32)
            ev_s[a:b] = spec["its_eig"].reshape(b - a, S, n)
            ts_s[a:b] = spec["its_ts"].reshape(b - a, S, n)
        tail = 50.0 * (1.0 - float(ci))                 # _its_alpha_tail_bounds :232-234
        ev, ev_ci, ts, ts_ci, rates, rate_ci = _posterior_summary(ev_s, ts_s, tail, 100.0 - tail)
        samples = {"eigenvalues": ev_s, "timescales": ts_s}
        dead = ~np.isfinite(ts).any(axis=1)
        if dead.any():
            # _its_fill_missing_timescales :403-419: lags whose samples gave no finite timescale take the
            # deterministic estimate from the regularised counts of that lag (_counts_for_lag)
            from .estimation import ensure_connected_counts

            for i in np.flatnonzero(dead):
                Ci = counts.view((k, k), offset_elems=int(i) * k * k).to_host().astype(float)
                ev[i], ts[i], rates[i] = deterministic_its_from_counts(
                    ensure_connected_counts(Ci, alpha=float(dirichlet_alpha)).counts, lags[i], n)
    res = ITSResult(lag_times=np.asarray(lags, dtype=int), eigenvalues=ev, eigenvalues_ci=ev_ci, timescales=ts,
                    timescales_ci=ts_ci, rates=rates, rates_ci=rate_ci)
    if plateau_m is not None and plateau_m >= 1 and L > 1:          # _its_optionally_attach_plateau :381-401
        win = detect_timescale_plateau(np.asarray(lags, float), ts, int(plateau_m), float(plateau_epsilon))
        if win is not None:
            dt = float(time_per_frame_ps or 1.0)
            res.recommended_lag_window = (win[0] * dt, win[1] * dt)
    if return_samples:
        res.samples = samples                                       # engine extra: the per-sample spectra
    return res
