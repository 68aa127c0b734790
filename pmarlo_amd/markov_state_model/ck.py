"""Chapman-Kolmogorov test on discrete trajectories: the microstate branch of
pmarlo.markov_state_model.ck_runner.run_ck (S/markov_state_model/ck_runner.py:135-153
preprocessing, :240-270 state selection, :155-176 test) with the count matrices at every lag
multiple from ONE lag-scan launch sequence and the matrix powers on the matrix cores.

Not mirrored: the macrostate branch (needs PCCA+), CSV / JSON / PNG
side outputs."""

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


def run_ck(dtrajs: Sequence[np.ndarray], lag_time: int, macro_k: int = 4, min_trans: int = 50,
           top_n_micro: int = 50, factors: Iterable[int] = (2, 3, 4, 5)) -> CKRunResult:
    """mse[f] = mean((T(tau)^f - T(f tau))^2) on the `top_n_micro` most populated connected
    microstates; factors whose lag-f*tau count rows do not all reach `min_trans` are listed in
    ``insufficient_k``.  ``macro_k`` is accepted for signature compatibility (macro branch not
    available)."""
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
    ctau = _device_counts(eng, trajs, active.size, [lag_time])
    if ctau is None:
        return res
    Ctau = ctau.to_host()[0]
    pops = Ctau.sum(axis=1) + Ctau.sum(axis=0)
    if np.count_nonzero(pops) == 0:
        return res
    top = np.argsort(-pops, kind="stable")[: min(int(top_n_micro), pops.size)]
    micro = _relabel(trajs, top, active.size)
    n_sel = int(top.size)
    lags = [lag_time] + [lag_time * f for f in factors_list]
    Cd = _device_counts(eng, micro, n_sel, lags)        # int64 [1 + F, n_sel, n_sel], stays on the device
    if Cd is None:
        return res
    nn = n_sel * n_sel
    rows = [eng.transition_matrix(Cd.view((n_sel, n_sel), np.int64, offset_elems=i * nn), mode=0)
            for i in range(len(lags))]
    rowsums = np.stack([r["rowsum"].to_host() for r in rows])
    if np.any(rowsums[0] < min_trans):
        return res
    usable = [i for i in range(len(factors_list)) if not np.any(rowsums[1 + i] < min_trans)]
    res.mode = "micro"
    res.selected_states = active[top]
    if not usable:
        return res
    Tk = eng.empty((len(usable), n_sel, n_sel), np.float64)
    for j, u in enumerate(usable):
        Tk.view((n_sel, n_sel), np.float64, offset_elems=j * nn).copy_from(rows[1 + u]["T"])
    mse, _ = eng.ck_test(rows[0]["T"], Tk, [factors_list[u] for u in usable])
    for j, u in enumerate(usable):
        res.mse[factors_list[u]] = float(mse[j])
        res.insufficient_k.remove(factors_list[u])
    return res
