"""Implied timescales: mirror of ITSMixin (S/markov_state_model/_its.py:137-192, 272-357,
543-668) and utils.safe_timescales (S/markov_state_model/utils.py:17-57).

The reference's default ITS is the median over 100 Bayesian posterior samples drawn by
deeptime's C++ sampler, which cannot be reproduced outside deeptime (SURVEY.md hard part 4).
This engine computes the DETERMINISTIC definition: _summarize_its_stats applied to the single
maximum-likelihood matrix T(tau) = rownorm(C(tau) + alpha) on the active set; confidence
intervals collapse onto the estimate."""

from __future__ import annotations

from typing import Sequence

import numpy as np

from ..device import get_engine
from .estimation import _concat_dtrajs
from .results import ITSResult

__all__ = ["safe_timescales", "compute_implied_timescales"]

EPS = 1e-12


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


def compute_implied_timescales(dtrajs: Sequence[np.ndarray], n_states: int, lag_times: Sequence[int] | None = None,
                               n_timescales: int = 5, *, alpha: float = 1e-3) -> ITSResult:
    """Lag scan on the device: batched counts for all lags, one packed transition matrix per
    lag, one batched spectrum solve."""
    lens = [len(d) for d in dtrajs if len(d)]
    if not lens:
        return ITSResult()
    max_valid = min(lens) - 1
    if lag_times is None:
        lag_times = [1, 2, 3, 5, 8, 10, 15, 20, 30, 40, 50, 75, 100, 150, 200]
    lags = [int(v) for v in lag_times if 1 <= int(v) <= max_valid]  # :476-500
    if not lags:
        return ITSResult()
    labels, segs = _concat_dtrajs(dtrajs, n_states)
    eng = get_engine()
    k, L, n = int(n_states), len(lags), int(n_timescales)
    starts = np.asarray([s for s, _ in segs], np.int64)
    stops = np.asarray([e for _, e in segs], np.int64)
    counts, _ = eng.count_transitions_lagscan(eng.to_device(labels), k, lags, starts=starts, stops=stops)
    Tb = eng.empty((L, k, k), np.float64)
    nb = eng.empty((L,), np.int32)
    from .._lib import check, lib

    for i in range(L):  # per-lag active set + alpha, packed in place (device to device)
        rows = eng.empty((k,), np.float64)
        act, inv = eng.empty((k,), np.int32), eng.empty((k,), np.int32)
        check(lib.msm_transition_matrix(eng.handle, counts.ptr + i * k * k * 8, 0, k, 1, float(alpha), 1e-12,
                                        Tb.ptr + i * k * k * 8, act.ptr, inv.ptr, nb.ptr + 4 * i, rows.ptr, None),
              eng.handle)
    spec = eng.spectrum(Tb, n=nb, n_its=n, lags=[float(v) for v in lags], want_pi=False)
    ev, ts = spec["its_eig"], spec["its_ts"]
    with np.errstate(divide="ignore", invalid="ignore"):
        rates = np.where(np.isfinite(ts), 1.0 / ts, np.nan)
    ci = lambda a: np.stack([a, a], axis=-1)  # noqa: E731  deterministic: zero-width intervals
    return ITSResult(lag_times=np.asarray(lags, dtype=int), eigenvalues=ev, eigenvalues_ci=ci(ev), timescales=ts,
                     timescales_ci=ci(ts), rates=rates, rates_ci=ci(rates))
