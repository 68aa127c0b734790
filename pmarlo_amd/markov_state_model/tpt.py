"""Transition-path theory and macrostate lumping on the GPU.

Mirrors the numerics behind pmarlo.markov_state_model._tpt.TPTMixin (S/markov_state_model/
_tpt.py:39-160 reactive_flux / compute_committor, :255-347 net / gross flux, rate, mfpt -- all
delegated to deeptime 0.4.5 in the reference; the published dense algorithm is restated, parity
unpinned) and lump_micro_to_macro_T / compute_macro_populations / compute_macro_mfpt
(S/markov_state_model/_msm_utils.py:103-160).  Not mirrored: pathway decomposition, flux
coarse-graining, PCCA+ itself."""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ..device import get_engine

__all__ = ["ReactiveFlux", "reactive_flux", "compute_committor", "lump_micro_to_macro_T", "compute_macro_populations",
           "compute_macro_mfpt"]


@dataclass
class ReactiveFlux:
    """The attributes of deeptime's ReactiveFlux that the reference reads."""

    source_states: list
    sink_states: list
    forward_committor: np.ndarray
    backward_committor: np.ndarray
    gross_flux: np.ndarray
    net_flux: np.ndarray
    total_flux: float
    rate: float
    mfpt: float
    stationary_distribution: np.ndarray


def _roles(n: int, source_states, sink_states) -> tuple[np.ndarray, list, list]:
    source = np.unique(np.asarray(source_states, dtype=int))
    sink = np.unique(np.asarray(sink_states, dtype=int))
    if source.size == 0 or sink.size == 0:
        raise ValueError("Source and sink must each contain at least one state")
    if len(np.intersect1d(source, sink)) > 0:
        raise ValueError(f"Source and sink states must not overlap. Source: {source.tolist()}, Sink: {sink.tolist()}")
    if source.min() < 0 or sink.min() < 0 or source.max() >= n or sink.max() >= n:
        raise ValueError("Source / sink state index out of range")
    role = np.zeros(n, dtype=np.int32)
    role[source] = 1
    role[sink] = 2
    return role, source.tolist(), sink.tolist()


def _check_T(T: np.ndarray) -> np.ndarray:
    T = np.ascontiguousarray(T, dtype=np.float64)
    if T.ndim != 2 or T.shape[0] != T.shape[1] or T.shape[0] < 2:
        raise ValueError("transition matrix must be square with at least two states")
    return T


def reactive_flux(transition_matrix, stationary_distribution, source_states, sink_states) -> ReactiveFlux:
    """Committors, gross / net flux, total flux, rate k_AB = F / sum_i pi_i q-_i and mfpt = 1 / k_AB."""
    if transition_matrix is None or stationary_distribution is None:
        raise ValueError("Must call build_msm() before computing reactive flux. "
                         "Transition matrix and stationary distribution are required.")
    T = _check_T(transition_matrix)
    pi = np.ascontiguousarray(stationary_distribution, dtype=np.float64)
    role, source, sink = _roles(T.shape[0], source_states, sink_states)
    eng = get_engine()
    out = eng.reactive_flux(eng.to_device(T), eng.to_device(pi), role)
    if np.any(out["info"] != 0):
        raise np.linalg.LinAlgError("committor system is singular (disconnected transition matrix?)")
    tot = out["totals"].to_host()
    return ReactiveFlux(source, sink, out["qplus"].to_host(), out["qminus"].to_host(), out["gross"].to_host(),
                        out["net"].to_host(), float(tot[0]), float(tot[2]), float(tot[3]), pi)


def compute_committor(transition_matrix, source_states, sink_states, forward: bool = True,
                      stationary_distribution=None) -> np.ndarray:
    """Forward committor q+ (probability of reaching the sink before the source), or the backward
    committor q- (needs the stationary distribution; computed on the device when not given)."""
    if transition_matrix is None:
        raise ValueError("Must call build_msm() before computing committors")
    T = _check_T(transition_matrix)
    role, _, _ = _roles(T.shape[0], source_states, sink_states)
    eng = get_engine()
    Td = eng.to_device(T)
    if stationary_distribution is None:
        if forward:
            pi_d = eng.to_device(np.full(T.shape[0], 1.0 / T.shape[0]))   # unused by the forward system
        else:
            pi_d = eng.spectrum(Td, n_its=0)["pi"].view((T.shape[0],))
    else:
        pi_d = eng.to_device(np.ascontiguousarray(stationary_distribution, dtype=np.float64))
    out = eng.reactive_flux(Td, pi_d, role, want_flux=False)
    if out["info"][0 if forward else 1] != 0:
        raise np.linalg.LinAlgError("committor system is singular (disconnected transition matrix?)")
    return (out["qplus"] if forward else out["qminus"]).to_host()


def compute_macro_populations(pi_micro: np.ndarray, micro_to_macro: np.ndarray) -> np.ndarray:
    """pi_macro[A] = sum_{i in A} pi_i, renormalised."""
    m = np.asarray(micro_to_macro, dtype=int)
    if m.size == 0:
        return np.zeros((0,), dtype=float)
    n_macro = int(m.max()) + 1
    eng = get_engine()
    n = m.size
    _, pm = eng.lump_macro(eng.to_device(np.eye(n)), eng.to_device(np.ascontiguousarray(pi_micro, np.float64)), m, n_macro)
    return pm.to_host()


def lump_micro_to_macro_T(T_micro: np.ndarray, pi_micro: np.ndarray, micro_to_macro: np.ndarray) -> np.ndarray:
    """T_macro[A, B] = F_AB / sum_B F_AB with the stationary flux F_AB = sum_{i in A, j in B} pi_i T_ij."""
    m = np.asarray(micro_to_macro, dtype=int)
    if m.size == 0:
        return np.zeros((0, 0), dtype=float)
    T = np.ascontiguousarray(T_micro, dtype=np.float64)
    eng = get_engine()
    Tm, _ = eng.lump_macro(eng.to_device(T), eng.to_device(np.ascontiguousarray(pi_micro, np.float64)), m, int(m.max()) + 1)
    return Tm.to_host()


def compute_macro_mfpt(T_macro: np.ndarray) -> np.ndarray:
    """mfpt[i, j] = mean first-passage time i -> j of the discrete chain ((I - Q_j) t = 1); a singular
    system gives NaN in that column, as the reference does."""
    T = np.ascontiguousarray(T_macro, dtype=np.float64)
    n = T.shape[0]
    if n < 2:
        return np.zeros((n, n), dtype=float)
    eng = get_engine()
    out, info = eng.macro_mfpt(eng.to_device(T))
    M = out.to_host()
    for j in np.nonzero(info)[0]:
        M[:, j] = np.nan
        M[j, j] = 0.0
    return M
