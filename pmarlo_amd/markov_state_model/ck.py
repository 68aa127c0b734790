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

__all__ = ["CKRunResult", "run_ck"]


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
