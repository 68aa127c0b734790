"""Transition-path theory and macrostate lumping on the GPU.

Mirrors the numerics behind pmarlo.markov_state_model._tpt.TPTMixin (S/markov_state_model/
_tpt.py:39-160 reactive_flux / compute_committor, :255-347 net / gross flux, rate, mfpt -- all
delegated to deeptime 0.4.5 in the reference; the published dense algorithm is restated, parity
unpinned) and lump_micro_to_macro_T / compute_macro_populations / compute_macro_mfpt
(S/markov_state_model/_msm_utils.py:103-160), pathway_decomposition / coarse_grain_flux /
identify_transition_state_ensemble / find_bottleneck_states (_tpt.py:162-429).  The dense solves and
matrix products run on the device; the path search of the decomposition is a graph walk over the k x k
net-flux matrix on the host (deeptime does the same in Python)."""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ..device import get_engine

__all__ = ["ReactiveFlux", "reactive_flux", "compute_committor", "lump_micro_to_macro_T", "compute_macro_populations",
           "compute_macro_mfpt", "pathway_decomposition", "coarse_grain_flux", "identify_transition_state_ensemble",
           "find_bottleneck_states"]


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


def _widest_path(F: np.ndarray, src: int, dst: int):
    """Path src -> dst that maximises its smallest edge (max-min Dijkstra; ties: lower state index first).
    Returns (path, capacity) or (None, 0.0) when dst cannot be reached over positive edges."""
    import heapq

    n = F.shape[0]
    width = np.zeros(n)
    width[src] = np.inf
    prev = np.full(n, -1, dtype=int)
    done = np.zeros(n, dtype=bool)
    heap = [(-np.inf, src)]
    while heap:
        w, i = heapq.heappop(heap)
        if done[i]:
            continue
        done[i] = True
        if i == dst:
            break
        row = F[i]
        for j in np.flatnonzero(row > 0):
            cand = min(-w, row[j])
            if not done[j] and cand > width[j]:
                width[j] = cand
                prev[j] = i
                heapq.heappush(heap, (-cand, int(j)))
    if not done[dst] or width[dst] <= 0:
        return None, 0.0
    path = [dst]
    while path[-1] != src:
        path.append(int(prev[path[-1]]))
    return path[::-1], float(width[dst])


def pathway_decomposition(transition_matrix, stationary_distribution, source_states, sink_states, fraction: float = 0.99,
                          maxiter: int = 10000):
    """Dominant reactive pathways A -> B: repeatedly take the path of the largest bottleneck through the
    net flux and remove its capacity, until `fraction` of the total flux is collected or `maxiter` paths
    were taken (TPTMixin.pathway_decomposition, S/markov_state_model/_tpt.py:162-211; deeptime's
    ReactiveFlux.pathways restated from Metzner, Schuette, Vanden-Eijnden, MMS 7 (2009)).
    Returns (paths as lists of state indices, capacities)."""
    if not 0.0 < fraction <= 1.0:
        raise ValueError("fraction must be in (0, 1]")
    flux = reactive_flux(transition_matrix, stationary_distribution, source_states, sink_states)
    n = flux.net_flux.shape[0]
    A, B = flux.source_states, flux.sink_states
    # virtual end states n (feeds A) and n + 1 (drains B) make the search single-source / single-sink
    G = np.zeros((n + 2, n + 2))
    G[:n, :n] = flux.net_flux
    out_A = flux.net_flux[A].sum(axis=1)
    in_B = flux.net_flux[:, B].sum(axis=0)
    G[n, A] = out_A
    G[B, n + 1] = in_B
    total = float(out_A.sum())
    paths, caps, got = [], [], 0.0
    while len(paths) < int(maxiter) and total > 0 and got < fraction * total * (1.0 - 1e-14):
        path, cap = _widest_path(G, n, n + 1)
        if path is None or cap <= 1e-14 * total:
            break
        for a, b in zip(path[:-1], path[1:]):
            G[a, b] = max(0.0, G[a, b] - cap)
        paths.append([int(v) for v in path[1:-1]])
        caps.append(cap)
        got += cap
    return paths, np.asarray(caps, dtype=float)


def coarse_grain_flux(transition_matrix, stationary_distribution, source_states, sink_states, sets):
    """Reactive flux between sets of states (TPTMixin.coarse_grain_flux, _tpt.py:213-253; deeptime's
    ReactiveFlux.coarse_grain restated): every user set is split into its parts inside the source, the
    intermediates and the sink (states no set names form one more set), ordered source parts, intermediate
    parts, sink parts; gross flux is summed over the blocks (two matrix products on the device), the net
    flux is its antisymmetric positive part, committors are pi-weighted means.
    Returns (list of sets, ReactiveFlux over the sets)."""
    flux = reactive_flux(transition_matrix, stationary_distribution, source_states, sink_states)
    n = flux.gross_flux.shape[0]
    A, B = set(flux.source_states), set(flux.sink_states)
    user = [set(int(v) for v in s) for s in sets]
    for s in user:
        if s and (min(s) < 0 or max(s) >= n):
            raise ValueError("set member out of range")
    named = set().union(*user) if user else set()
    if len(named) != sum(len(s) for s in user):
        raise ValueError("sets must be disjoint")
    rest = set(range(n)) - named
    if rest:
        user.append(rest)
    inter = set(range(n)) - A - B
    parts_A = [s & A for s in user if s & A]
    parts_I = [s & inter for s in user if s & inter]
    parts_B = [s & B for s in user if s & B]
    tpt_sets = parts_A + parts_I + parts_B
    m = len(tpt_sets)
    S = np.zeros((n, m))
    for q, s in enumerate(tpt_sets):
        S[sorted(s), q] = 1.0
    eng = get_engine()
    Sd = eng.to_device(S)
    gross = eng.gemm(eng.to_device(np.ascontiguousarray(S.T)), eng.gemm(eng.to_device(flux.gross_flux), Sd)).to_host()
    np.fill_diagonal(gross, 0.0)
    net = np.maximum(gross - gross.T, 0.0)
    pi = flux.stationary_distribution
    pops = S.T @ pi
    with np.errstate(divide="ignore", invalid="ignore"):
        qp = np.where(pops > 0, (S.T @ (pi * flux.forward_committor)) / pops, 0.0)
        qm = np.where(pops > 0, (S.T @ (pi * flux.backward_committor)) / pops, 0.0)
    a_idx = list(range(len(parts_A)))
    b_idx = list(range(len(parts_A) + len(parts_I), m))
    total = float(net[a_idx].sum())
    z = float(np.sum(pops * qm))
    rate = total / z if z > 0 else float("nan")
    cg = ReactiveFlux(a_idx, b_idx, qp, qm, gross, net, total, rate, 1.0 / rate if rate > 0 else float("inf"), pops)
    return tpt_sets, cg


def identify_transition_state_ensemble(transition_matrix, source_states, sink_states, tolerance: float = 0.1) -> np.ndarray:
    """States whose forward committor lies within `tolerance` of one half (_tpt.py:349-385)."""
    q = compute_committor(transition_matrix, source_states, sink_states, forward=True)
    return np.where((q >= 0.5 - tolerance) & (q <= 0.5 + tolerance))[0]


def find_bottleneck_states(transition_matrix, stationary_distribution, source_states, sink_states, top_n: int = 10) -> np.ndarray:
    """The top_n states by reactive flux passing through them, (row sum + column sum) / 2 of the gross flux,
    descending (_tpt.py:387-426)."""
    flux = reactive_flux(transition_matrix, stationary_distribution, source_states, sink_states)
    through = 0.5 * (flux.gross_flux.sum(axis=1) + flux.gross_flux.sum(axis=0))
    return np.argsort(through)[::-1][:int(top_n)]
