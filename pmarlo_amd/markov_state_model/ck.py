"""Chapman-Kolmogorov test on discrete trajectories: the microstate branch of
pmarlo.markov_state_model.ck_runner.run_ck (S/markov_state_model/ck_runner.py:135-153
preprocessing, :240-270 state selection, :155-176 test) with the count matrices at every lag
multiple from ONE lag-scan launch sequence and the matrix powers on the matrix cores.

The macrostate branch (:178-213) runs first, as in the reference: PCCA+ (pcca.py) on the lag-1 matrix of
the connected microstates; deeptime's PCCA+ only accepts matrices with detailed balance, so raw counts
normally fall through to the microstate branch.  Not mirrored: CSV / JSON / PNG side outputs."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Sequence

import numpy as np

from ..device import get_engine

__all__ = ["CKRunResult", "run_ck", "CKTestResult", "compute_ck_test_micro", "compute_ck_test_macrostates",
           "select_lag_time_ck"]


@dataclass
class CKRunResult:
    mse: Dict[int, float] = field(default_factory=dict)
    mode: str = "none"
    insufficient_k: List[int] = field(default_factory=list)
    selected_states: np.ndarray | None = None
    macro_labels: np.ndarray | None = None     # macro mode: macrostate of every selected (connected) microstate

    @property
    def max_error(self) -> float:
        return float(np.sqrt(max(self.mse.values()))) if self.mse else float("inf")


def _validate(dtrajs: Sequence[np.ndarray], lag_time: int, factors: Sequence[int]) -> None:
    if not dtrajs:
        raise ValueError("No trajectories provided for analysis")
    if lag_time <= 0:
        raise ValueError(f"Lag time must be positive, got {lag_time}")
    if not factors:
        raise ValueError("No lag factors provided for analysis")
    bad = [f for f in factors if f <= 1]
    if bad:
        raise ValueError(f"All lag factors must be > 1, got {bad}")


def _relabel(trajs: Sequence[np.ndarray], keep: np.ndarray, n_old: int) -> list[np.ndarray]:
    """Renumber the kept states 0..len(keep)-1 and DROP the frames of all other states (the
    reference filters the sequences, so later transitions bridge the removed frames)."""
    lut = -np.ones(n_old, dtype=np.int64)
    lut[keep] = np.arange(len(keep))
    out = []
    for t in trajs:
        m = lut[np.asarray(t, dtype=np.int64)]
        out.append(m[m >= 0].astype(np.int32))
    return out


def _device_counts(eng, trajs: Sequence[np.ndarray], n_states: int, lags: Sequence[int]):
    """Count matrices int64 [len(lags), n, n] ON THE DEVICE of a list of label sequences (pairs
    never cross a sequence boundary), or None when there are no frames."""
    lens = np.asarray([len(t) for t in trajs], dtype=np.int64)
    stops = np.cumsum(lens)
    starts = stops - lens
    if int(lens.sum()) == 0:
        return None
    labels = np.concatenate([np.asarray(t, dtype=np.int32) for t in trajs])
    keep = lens > 0
    counts, _ = eng.count_transitions_lagscan(eng.to_device(labels), n_states, list(lags), starts=starts[keep],
                                              stops=stops[keep])
    return counts


def _ck_on_trajs(eng, trajs: Sequence[np.ndarray], n_sel: int, lag_time: int, factors_list: List[int], min_trans: int,
                 res: CKRunResult) -> bool:
    """_ck_on_trajs (:156-175) preceded by the caller's row-count check on the lag-tau matrix: False when a
    lag-tau row has fewer than min_trans counts (nothing is written), True otherwise with res.mse /
    res.insufficient_k updated."""
    lags = [lag_time] + [lag_time * f for f in factors_list]
    Cd = _device_counts(eng, trajs, n_sel, lags)        # int64 [1 + F, n_sel, n_sel], stays on the device
    if Cd is None:
        return False
    nn = n_sel * n_sel
    rows = [eng.transition_matrix(Cd.view((n_sel, n_sel), np.int64, offset_elems=i * nn), mode=0)
            for i in range(len(lags))]
    rowsums = np.stack([r["rowsum"].to_host() for r in rows])
    if np.any(rowsums[0] < min_trans):
        return False
    usable = [i for i in range(len(factors_list)) if not np.any(rowsums[1 + i] < min_trans)]
    if not usable:
        return True
    Tk = eng.empty((len(usable), n_sel, n_sel), np.float64)
    for j, u in enumerate(usable):
        Tk.view((n_sel, n_sel), np.float64, offset_elems=j * nn).copy_from(rows[1 + u]["T"])
    mse, _ = eng.ck_test(rows[0]["T"], Tk, [factors_list[u] for u in usable])
    for j, u in enumerate(usable):
        res.mse[factors_list[u]] = float(mse[j])
        res.insufficient_k.remove(factors_list[u])
    return True


def _macro_labels(eng, trajs: Sequence[np.ndarray], n_micro: int, macro_k: int):
    """_attempt_macro_analysis (:178-199): PCCA+ on the lag-1 matrix of the connected microstates when
    there are more microstates than macrostates and the eigenvalue gap after macro_k is >= 0.01."""
    from .pcca import pcca_like_macrostates

    if n_micro <= macro_k:
        return None
    c1 = _device_counts(eng, trajs, n_micro, [1])
    if c1 is None:
        return None
    T1 = eng.transition_matrix(c1.view((n_micro, n_micro), np.int64), mode=0)["T"]
    spec = eng.spectrum(T1, n_its=0, n_watch=min(n_micro, macro_k + 1), want_pi=False, allow_unconverged=True)
    ev = np.sort(np.real(spec["ritz"][0][:min(n_micro, spec["p"])]))[::-1]        # _eigen_gap :99-109
    if ev.size <= macro_k or float(ev[macro_k - 1] - ev[macro_k]) < 0.01:
        return None
    return pcca_like_macrostates(T1.to_host(), n_macrostates=int(macro_k))


def run_ck(dtrajs: Sequence[np.ndarray], lag_time: int, macro_k: int = 4, min_trans: int = 50,
           top_n_micro: int = 50, factors: Iterable[int] = (2, 3, 4, 5)) -> CKRunResult:
    """mse[f] = mean((T(tau)^f - T(f tau))^2), on the PCCA+ macrostates of the connected microstates when
    PCCA+ accepts their lag-1 matrix (``mode == "macro"``), otherwise on the `top_n_micro` most populated
    microstates (``"micro"``); factors whose lag-f*tau count rows do not all reach `min_trans` are listed
    in ``insufficient_k``."""
    factors_list = [int(f) for f in factors if int(f) > 1]
    _validate(dtrajs, lag_time, factors_list)
    eng = get_engine()
    res = CKRunResult(insufficient_k=list(factors_list))
    n_states = int(max(int(np.max(t)) for t in dtrajs if len(t)) + 1)
    c1 = _device_counts(eng, dtrajs, n_states, [1])
    if c1 is None:
        return res
    C1 = c1.to_host()[0]
    active = np.where(C1.sum(axis=1) + C1.sum(axis=0) > 0)[0]
    if active.size == 0:
        return res
    trajs = _relabel(dtrajs, active, n_states)
    macro = _macro_labels(eng, trajs, int(active.size), int(macro_k))
    if macro is not None:
        n_macro = int(np.max(macro)) + 1
        macro_trajs = [macro[np.asarray(t, dtype=np.int64)].astype(np.int32) for t in trajs]
        if _ck_on_trajs(eng, macro_trajs, n_macro, lag_time, factors_list, min_trans, res):
            res.mode = "macro"
            res.selected_states = active
            res.macro_labels = macro
            return res
    ctau = _device_counts(eng, trajs, active.size, [lag_time])
    if ctau is None:
        return res
    Ctau = ctau.to_host()[0]
    pops = Ctau.sum(axis=1) + Ctau.sum(axis=0)
    if np.count_nonzero(pops) == 0:
        return res
    top = np.argsort(-pops, kind="stable")[: min(int(top_n_micro), pops.size)]
    micro = _relabel(trajs, top, active.size)
    if _ck_on_trajs(eng, micro, int(top.size), lag_time, factors_list, min_trans, res):
        res.mode = "micro"
        res.selected_states = active[top]
    return res


# ---- CKMixin (S/markov_state_model/_ck.py:61-228, helpers :258-358) as functions on label sequences -----------------
@dataclass
class CKTestResult:
    mse: Dict[int, float] = field(default_factory=dict)
    mode: str = "micro"
    insufficient_data: bool = False
    thresholds: Dict[str, int] = field(default_factory=dict)

    def to_dict(self):
        return {"mse": {int(k): float(v) for k, v in self.mse.items()}, "mode": self.mode,
                "insufficient_data": self.insufficient_data, "thresholds": self.thresholds}


def _factors(factors) -> List[int]:
    return [2, 3, 4, 5] if factors is None else [int(f) for f in factors if int(f) > 1]


def _ck_mse_chain(eng, trajs, n_sel: int, lag_time: int, factors: List[int], min_transitions: int, res: CKTestResult) -> None:
    """mse[f] for the factors in order; the first lag multiple with a row below min_transitions marks the result
    as insufficient and ends the scan (compute_ck_test_micro :93-108, _macrostates :143-154)."""
    lags = [lag_time] + [lag_time * f for f in factors]
    Cd = _device_counts(eng, trajs, n_sel, lags)
    if Cd is None:
        res.insufficient_data = True
        return
    nn = n_sel * n_sel
    rows = [eng.transition_matrix(Cd.view((n_sel, n_sel), np.int64, offset_elems=i * nn), mode=0) for i in range(len(lags))]
    rowsums = np.stack([r["rowsum"].to_host() for r in rows])
    if np.any(rowsums[0] < min_transitions):
        res.insufficient_data = True
        return
    good = 0
    while good < len(factors) and not np.any(rowsums[1 + good] < min_transitions):
        good += 1
    if good:
        Tk = eng.empty((good, n_sel, n_sel), np.float64)
        for j in range(good):
            Tk.view((n_sel, n_sel), np.float64, offset_elems=j * nn).copy_from(rows[1 + j]["T"])
        mse, _ = eng.ck_test(rows[0]["T"], Tk, factors[:good])
        for j in range(good):
            res.mse[factors[j]] = float(mse[j])
    if good < len(factors):
        res.insufficient_data = True


def compute_ck_test_micro(dtrajs: Sequence[np.ndarray], n_states: int, lag_time: int, factors=None, max_states: int = 50,
                          min_transitions: int = 5) -> CKTestResult:
    """CK test on the largest connected set of microstates (undirected count graph at lag_time), capped to the
    `max_states` most visited; frames of other states are dropped from the sequences."""
    from scipy.sparse.csgraph import connected_components

    factors = _factors(factors)
    res = CKTestResult(mode="micro", thresholds={"min_transitions_per_state": int(min_transitions), "max_states": int(max_states)})
    if not dtrajs or n_states <= 1 or lag_time <= 0:
        res.insufficient_data = True
        return res
    eng = get_engine()
    c = _device_counts(eng, dtrajs, int(n_states), [int(lag_time)])
    if c is None:
        res.insufficient_data = True
        return res
    C = c.to_host()[0].astype(float)
    _, lab = connected_components(((C + C.T) > 0).astype(int), directed=False, return_labels=True)
    idx = np.where(lab == int(np.argmax(np.bincount(lab))))[0]
    if idx.size > max_states:
        tot = (C + C.T).sum(axis=1)
        idx = idx[np.argsort(tot[idx])[::-1]][:max_states]
    if idx.size == 0:
        res.insufficient_data = True
        return res
    _ck_mse_chain(eng, _relabel(dtrajs, idx, int(n_states)), int(idx.size), int(lag_time), factors, int(min_transitions), res)
    return res


def compute_ck_test_macrostates(dtrajs: Sequence[np.ndarray], n_states: int, lag_time: int, macro_labels,
                                factors=None, min_transitions: int = 5, transition_matrix=None) -> CKTestResult:
    """CK test on macrostate trajectories (micro label -> macro_labels[label]); needs an eigenvalue gap
    lambda_2 - lambda_3 > 0.01 of the microstate matrix (`transition_matrix`, or the row-normalised lag counts)."""
    factors = _factors(factors)
    eng = get_engine()
    if transition_matrix is not None:
        Td = eng.to_device(np.ascontiguousarray(transition_matrix, dtype=np.float64))
    else:
        c = _device_counts(eng, dtrajs, int(n_states), [int(lag_time)]) if dtrajs and n_states > 0 and lag_time > 0 else None
        Td = None if c is None else eng.transition_matrix(c.view((int(n_states), int(n_states)), np.int64), mode=0)["T"]
    gap = None
    if Td is not None and Td.shape[0] > 2:
        spec = eng.spectrum(Td, n_its=0, n_watch=3, want_pi=False, allow_unconverged=True)
        ev = np.sort(np.real(spec["ritz"][0][:min(Td.shape[0], spec["p"])]))[::-1]
        gap = float(ev[1] - ev[2])
    if gap is None or gap <= 0.01:
        raise RuntimeError("Insufficient spectral gap for macrostate CK test")
    res = CKTestResult(mode="macro", thresholds={"min_transitions_per_state": int(min_transitions)})
    if not dtrajs or n_states <= 0 or lag_time <= 0:
        res.insufficient_data = True
        return res
    if macro_labels is None:
        raise RuntimeError("Macrostate labels are required for macrostate CK test")
    macro = np.asarray(macro_labels, dtype=np.int64)
    n_macro = int(macro.max() + 1)
    if n_macro <= 1:
        raise RuntimeError("Macrostate CK test requires at least two macrostates")
    mtraj = [macro[np.asarray(t, dtype=np.int64)].astype(np.int32) for t in dtrajs]
    _ck_mse_chain(eng, mtraj, n_macro, int(lag_time), factors, int(min_transitions), res)
    return res


def select_lag_time_ck(dtrajs: Sequence[np.ndarray], n_states: int, tau_candidates: Sequence[int], factor: int = 2,
                       mse_epsilon: float = 0.05) -> int:
    """The candidate with the smallest CK error mean((T(tau)^factor - T(factor tau))^2) over all microstates, tau = 2
    preferred over tau = 1 on a tie (select_lag_time_ck :157-171: the prefix rule it evaluates first is
    overwritten by the best-MSE rule, which is kept).  Counts for every tau and factor * tau come from one
    lag-scan pass."""
    taus = [int(t) for t in tau_candidates]
    if not taus:
        raise ValueError("no lag candidates")
    eng = get_engine()
    k = int(n_states)
    lags = sorted({t for t in taus} | {t * int(factor) for t in taus})
    Cd = _device_counts(eng, dtrajs, k, lags)
    if Cd is None:
        raise ValueError("no frames")
    where = {lv: i for i, lv in enumerate(lags)}
    T = {lv: eng.transition_matrix(Cd.view((k, k), np.int64, offset_elems=where[lv] * k * k), mode=0)["T"] for lv in lags}
    mses = []
    for t in taus:
        mse, _ = eng.ck_test(T[t], T[t * int(factor)].view((1, k, k)), [int(factor)])
        mses.append(float(mse[0]))
    best = taus[int(np.nanargmin(mses))]
    if best == 1 and 2 in taus and mses[taus.index(2)] <= mses[int(np.nanargmin(mses))] + 1e-12:
        best = 2
    return int(best)
