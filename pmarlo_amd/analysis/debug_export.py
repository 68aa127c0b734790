"""Pre-build dataset diagnostics: mirror of pmarlo.analysis.debug_export.compute_analysis_debug
(S/analysis/debug_export.py:28-201, helpers :354-586) and of pmarlo.utils.scc (S/utils/scc.py).

The passes over the frames run on the device: lag-tau transition counts (sliding or strided; pairs with an
unassigned frame are skipped, pairs may bridge one), state visits, dwell-time runs.  Connectivity of the
k x k count graph (scipy, as the reference), medians over the list of runs and the summary itself are host
logic.  Pinned by tests/golden/debug.json, made by importing the reference.  The export_* writers of the
reference module (files on disk) are not mirrored."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Iterable, List, Mapping, Sequence

import numpy as np

from ..device import get_engine
from .counting import expected_pairs

__all__ = ["AnalysisDebugData", "SCCSummary", "analyse_scc", "compute_component_coverage", "compute_analysis_debug"]


@dataclass
class AnalysisDebugData:
    summary: Dict[str, Any]
    counts: np.ndarray
    state_counts: np.ndarray
    component_labels: np.ndarray

    def to_summary_dict(self) -> Dict[str, Any]:
        payload = dict(self.summary)
        payload["component_labels"] = self.component_labels.astype(int).tolist()
        payload["state_counts"] = self.state_counts.astype(float).tolist()
        payload["counts_nonzero"] = int(np.count_nonzero(self.counts))
        payload["counts_density"] = float(payload["counts_nonzero"] / float(self.counts.size)) if self.counts.size else 0.0
        return payload


@dataclass(frozen=True)
class SCCSummary:
    n_nodes: int
    component_labels: np.ndarray
    components: list
    component_sizes: np.ndarray
    largest_component: np.ndarray
    largest_fraction: float | None
    state_indices: np.ndarray


def analyse_scc(counts: np.ndarray, *, state_indices: Iterable[int] | None = None) -> SCCSummary:
    """Strongly connected components of the graph with an edge wherever counts > 0 (S/utils/scc.py:66-133)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components

    counts = np.asarray(counts)
    if counts.ndim != 2 or counts.shape[0] != counts.shape[1]:
        raise ValueError("count matrix must be square")
    n = int(counts.shape[0])
    idx = np.arange(n, dtype=int) if state_indices is None else np.asarray(list(state_indices), dtype=int).reshape(n)
    empty = np.empty((0,), dtype=int)
    if n == 0:
        return SCCSummary(0, empty, [], empty, empty, None, empty)
    n_comp, labels = connected_components(csgraph=csr_matrix(counts > 0.0), directed=True, connection="strong")
    comps = [np.flatnonzero(labels == c).astype(int) for c in range(n_comp)]
    comps = [c for c in comps if c.size]
    sizes = np.asarray([c.size for c in comps], dtype=int)
    if sizes.size:
        top = int(np.argmax(sizes))
        largest, frac = comps[top], float(sizes[top] / n)
    else:
        largest, frac = empty, None
    return SCCSummary(n, np.asarray(labels, dtype=int), comps, sizes, largest, frac, idx)


def compute_component_coverage(population: np.ndarray, component_indices: Sequence[int]) -> float | None:
    population = np.asarray(population)
    if population.size == 0:
        return None
    total = float(np.sum(population))
    if total <= 0.0:
        return None
    return float(np.sum(population[np.asarray(component_indices, dtype=int)])) / total


def _coerce_dtrajs(dtrajs) -> List[np.ndarray]:
    out: List[np.ndarray] = []
    for traj in (dtrajs.values() if isinstance(dtrajs, Mapping) else dtrajs):
        try:
            out.append(np.asarray(traj, dtype=int).reshape(-1))
        except Exception:
            continue
    return out


def _valid_segment_lengths(dtrajs: Sequence[np.ndarray]) -> List[int]:
    """Lengths of the maximal stretches of assigned (>= 0) frames, trajectory by trajectory (:428-444)."""
    out: List[int] = []
    for t in dtrajs:
        if t.size == 0:
            continue
        ok = np.concatenate([[False], t >= 0, [False]])
        edges = np.flatnonzero(ok[1:] != ok[:-1])
        out.extend(int(b - a) for a, b in zip(edges[::2], edges[1::2]))
    return out


def _dwell_statistics(eng, dtrajs: Sequence[np.ndarray], n_states: int) -> Dict[str, Any]:
    """Per-state min / max / mean / median run length and number of runs; unassigned frames are removed from a
    trajectory before its runs are taken, as the reference does (:447-530)."""
    parts: List[np.ndarray] = []
    for t in dtrajs:
        v = t[t >= 0] if np.any(t < 0) else t
        if v.size:
            parts.append(v.astype(np.int32))
            parts.append(np.asarray([-1], dtype=np.int32))          # separator: runs never cross trajectories
    zeros_i, zeros_f = [0] * n_states, [0.0] * n_states
    if not parts:
        return {"per_state_dwell_min": zeros_i, "per_state_dwell_max": list(zeros_i), "per_state_dwell_mean": zeros_f,
                "per_state_dwell_median": list(zeros_f), "per_state_transition_counts": list(zeros_i)}
    stats, run_state, run_len = eng.run_lengths(eng.to_device(np.concatenate(parts)), n_states)
    number = stats[3]
    seen = number > 0
    order = np.lexsort((run_len, run_state))
    sorted_len = run_len[order].astype(np.float64)
    first = np.concatenate([[0], np.cumsum(np.bincount(run_state, minlength=n_states))])[:-1]
    median = np.zeros(n_states)
    for s in np.flatnonzero(seen):
        m, a = int(number[s]), int(first[s])
        median[s] = sorted_len[a + m // 2] if m % 2 else 0.5 * (sorted_len[a + m // 2 - 1] + sorted_len[a + m // 2])
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = np.where(seen, stats[2] / np.maximum(number, 1), 0.0)
    return {"per_state_dwell_min": [int(v) if ok else 0 for v, ok in zip(stats[0], seen)],
            "per_state_dwell_max": [int(v) for v in stats[1]],
            "per_state_dwell_mean": [float(v) for v in mean],
            "per_state_dwell_median": [float(v) for v in median],
            "per_state_transition_counts": [int(v) for v in number]}


def compute_analysis_debug(dataset: Mapping[str, Any], *, lag: int, count_mode: str = "sliding") -> AnalysisDebugData:
    """Counts, visits, connectivity, dwell times and warnings of a discretised dataset ahead of the MSM build."""
    dtrajs = _coerce_dtrajs(dataset.get("dtrajs", ()))
    declared = sum(int(d.size) for d in dtrajs)
    if not dtrajs or all(d.size == 0 for d in dtrajs):
        raise ValueError(
            "Cannot compute analysis debug statistics: dataset has no discrete trajectories (dtrajs). "
            "The dataset must be discretized (clustered) before transition counts can be computed. "
            f"Dataset contains {declared} assigned frames, "
            "but no state assignments are present. Run discretization first.")
    top = max((int(np.max(d)) for d in dtrajs if d.size), default=-1)
    n_states = top + 1 if top >= 0 else 0
    if n_states == 0:
        raise ValueError(
            "Cannot compute MSM statistics: no valid states detected in discrete trajectories. "
            f"Dataset contains {declared} assigned frames, "
            f"but all discrete trajectory values are negative or empty. "
            "This may indicate clustering failed or produced invalid state assignments. "
            "Check your clustering configuration and ensure valid state labels are generated.")
    eng = get_engine()
    lag = int(lag)
    stride = lag if str(count_mode).lower() == "strided" else 1
    lens = np.asarray([d.size for d in dtrajs], dtype=np.int64)
    stops = np.cumsum(lens)
    starts = stops - lens
    labels = eng.to_device(np.concatenate(dtrajs).astype(np.int32))
    if lag > 0:
        keep = lens > 0
        cd, pd_ = eng.count_transitions(labels, n_states, lag, starts=starts[keep], stops=stops[keep], stride=max(1, stride))
        counts, total_pairs = cd.to_host().astype(float), int(pd_.to_host()[0])
    else:
        counts, total_pairs = np.zeros((n_states, n_states)), 0
    state_counts = eng.state_counts(labels, n_states).to_host().astype(int)
    valid_lengths = _valid_segment_lengths(dtrajs)
    row_sums = counts.sum(axis=1)
    zero_rows = int(np.sum(np.isclose(row_sums, 0.0)))
    scc = analyse_scc(counts)
    components = [c.astype(int).tolist() for c in scc.components]
    labels_c = scc.component_labels.astype(int)
    largest = scc.largest_component.astype(int).tolist()
    cover = compute_component_coverage(state_counts, largest)
    with np.errstate(divide="ignore", invalid="ignore"):
        diag = float(np.trace(counts / np.where(row_sums == 0.0, 1.0, row_sums)[:, None]) / n_states)
    isolated: List[int] = []
    if len(components) > 1:
        big = max(range(len(components)), key=lambda i: len(components[i]))
        isolated = [int(s) for s in range(len(labels_c)) if labels_c[s] != big]
    warnings: List[Dict[str, Any]] = []
    if total_pairs < 5000:
        warnings.append({"code": "TOTAL_PAIRS_LT_5000",
                         "message": f"Too few (t, t+tau) pairs for reliable MSM (observed {total_pairs}, requires >=5000)."})
    if zero_rows > 0:
        warnings.append({"code": "ZERO_ROW_STATES_PRESENT",
                         "message": "States with zero outgoing counts detected before regularisation; "
                                    "prune states or lower lag to avoid singular rows."})
    if cover is not None and cover < 0.9:
        warnings.append({"code": "SCC_COVERAGE_LT_0.90",
                         "message": f"Largest strongly connected component covers only {cover:.2%} of visited frames."})
    stride_pairs = 1 if count_mode == "sliding" else max(1, lag)
    expected = expected_pairs(valid_lengths, lag, stride_pairs)
    if abs(expected - total_pairs) > len(valid_lengths):
        raise ValueError(
            f"Pair counting mismatch: counted {total_pairs} pairs but expected "
            f"{expected} pairs based on actual dtraj lengths (tolerance: {len(valid_lengths)}). "
            f"Dtraj lengths: {[int(v) for v in lens]}, lag: {lag}, stride: {stride_pairs}. "
            f"This indicates a bug in the transition counting logic.")
    order = np.argsort(state_counts)
    low = order[:min(10, n_states)].tolist()
    summary: Dict[str, Any] = {
        "tau_frames": lag, "count_mode": str(count_mode), "total_frames_declared": int(lens.sum()),
        "total_frames_with_states": int(state_counts.sum()), "total_pairs": int(total_pairs),
        "counts_shape": [n_states, n_states], "zero_rows": zero_rows,
        "states_observed": int(np.count_nonzero(state_counts)), "largest_scc_size": int(scc.largest_component.size),
        "largest_scc_frame_fraction": float(cover) if cover is not None else None,
        "component_sizes": [len(c) for c in components], "n_components": len(components),
        "is_fully_connected": len(components) == 1, "isolated_states": isolated, "stride": int(stride_pairs),
        "expected_pairs": int(expected), "counted_pairs": int(total_pairs), "total_pairs_predicted": int(expected),
        "diag_mass": diag, "warnings": warnings, "dwell_time_stats": _dwell_statistics(eng, dtrajs, n_states),
        "occupancy_tail": {"lowest_occupancy_states": low, "lowest_occupancy_counts": state_counts[low].tolist()},
    }
    return AnalysisDebugData(summary=summary, counts=counts, state_counts=state_counts, component_labels=labels_c)
