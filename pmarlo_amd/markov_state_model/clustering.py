"""Microstate clustering: mirror of pmarlo.markov_state_model.clustering
(S/markov_state_model/clustering.py:43-90 ClusteringResult, :395-665 cluster_microstates,
:364-392 _remap_labels_and_compute_inertia)."""

from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Any, Literal

import numpy as np

from ..device import get_engine
from ..pipeline import MSMPipeline

logger = logging.getLogger("pmarlo")

__all__ = ["ClusteringResult", "cluster_microstates", "silhouette_score"]

_ALLOWED_KW = {"max_iter", "tolerance", "n_init", "init_centers"}


@dataclass
class ClusteringResult:
    labels: np.ndarray
    n_states: int
    rationale: str | None = None
    centers: np.ndarray | None = None

    @property
    def output_shape(self) -> tuple[int, ...]:
        return (self.n_states,)


def _fit_once(pipe, yd, k, seed, max_iter, tol, init):
    eng = pipe.eng
    md = eng.empty((yd.shape[0],), np.float64)
    centers0 = eng.to_device(np.ascontiguousarray(init, np.float64)) if init is not None else None
    labels, centers, _ = pipe.cluster(yd, k, seed=seed, max_iter=max_iter, tol=tol, centers=centers0, mindist=md)
    inertia = float(eng.sum_f64(md).to_host()[0])
    return labels, centers, inertia


def silhouette_score(X: np.ndarray, labels: np.ndarray) -> float:
    """sklearn.metrics.silhouette_score (Euclidean) on the GPU: O(n^2 d) pair distances."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    labels = np.asarray(labels)
    uniq, dense = np.unique(labels, return_inverse=True)
    if not 2 <= uniq.size <= X.shape[0] - 1:
        raise ValueError(f"Number of labels is {uniq.size}. Valid values are 2 to n_samples - 1 (inclusive)")
    if uniq.size > 32:
        raise NotImplementedError("silhouette_score on the device supports at most 32 clusters")
    order = np.argsort(dense, kind="stable")
    offsets = np.concatenate([[0], np.cumsum(np.bincount(dense, minlength=uniq.size))])
    eng = get_engine()
    score, _ = eng.silhouette(eng.to_device(X[order]), offsets)
    return score


_MAX_SILHOUETTE_POINTS = 200_000


def _auto_select_n_states(Y, random_state, *, sample_size, override_n_states, kwargs) -> tuple[int, str]:
    """k = 4..20 scored by the silhouette coefficient of a k-means fit on (a random subset of) the data
    (clustering.py:156-233); the best score wins, the first k on ties."""
    if override_n_states is not None:
        if override_n_states <= 0:
            raise ValueError(f"override_n_states must be a positive integer; received {override_n_states}.")
        return int(override_n_states), f"auto-override={override_n_states}"
    note = ""
    Ys = Y
    if sample_size is not None:
        if sample_size <= 1:
            raise ValueError("sample_size must be greater than 1 when sampling for silhouette scoring.")
        eff = min(int(sample_size), int(Y.shape[0]))
        idx = np.random.default_rng(random_state).choice(Y.shape[0], size=eff, replace=False)
        Ys = Y[idx]
        note = f" sample={eff}"
    if Ys.shape[0] > _MAX_SILHOUETTE_POINTS:
        raise ValueError(f"silhouette scoring is O(n^2): pass silhouette_sample_size (got {Ys.shape[0]} points)")
    eng = get_engine()
    pipe = MSMPipeline(eng)
    yd = eng.to_device(np.ascontiguousarray(Ys))
    Ys64 = np.ascontiguousarray(Ys, dtype=np.float64)
    seed = 0 if random_state is None else int(random_state)
    scores = []
    for k in range(4, 21):
        if Ys.shape[0] <= k:
            break
        # the engine's seeded random-point start has no k-means++ spreading: a few restarts (lowest
        # inertia wins) keep a poor start from deciding the scan
        best = None
        for r in range(max(3, int(kwargs.get("n_init", 1)))):
            labels, _, inertia = _fit_once(pipe, yd, k, seed + 7919 * r, int(kwargs.get("max_iter", 100)),
                                           float(kwargs.get("tolerance", 1e-5)), None)
            if best is None or inertia < best[1]:
                best = (labels, inertia)
        lab = best[0].to_host()
        scores.append((k, silhouette_score(Ys64, lab) if np.unique(lab).size > 1 else -1.0))
    if not scores:
        raise ValueError("too few samples for the n_states='auto' scan (needs more than 4)")
    chosen, best = max(scores, key=lambda t: t[1])
    logger.info("Auto-selected %d states with silhouette score %.3f", chosen, best)
    return int(chosen), f"silhouette={best:.3f}{note}"


def cluster_microstates(
    Y: np.ndarray,
    method: Literal["auto", "minibatchkmeans", "kmeans"] = "auto",
    n_states: int | Literal["auto"] = 100,
    random_state: int | None = 42,
    minibatch_threshold: int = 5_000_000,
    *,
    silhouette_sample_size: int | None = None,
    auto_n_states_override: int | None = None,
    **kwargs: Any,
) -> ClusteringResult:
    """k-means microstates on the GPU.  ``method`` is accepted for API compatibility: on this
    engine every size runs full-batch Lloyd (the reference switches to mini-batch above
    ``minibatch_threshold`` only to bound CPU time).  Labels are densified and centres
    recomputed as member means exactly as the reference does after its estimator returns."""
    Y = np.asarray(Y)
    if Y.ndim != 2:
        raise ValueError(f"Input must be 2D array, got shape {Y.shape}")
    if Y.shape[1] == 0:
        raise ValueError("Input array must have at least one feature")
    if Y.shape[0] == 0:
        return ClusteringResult(labels=np.zeros((0,), dtype=int), n_states=0, rationale="empty input", centers=None)
    if method not in ("auto", "minibatchkmeans", "kmeans"):
        raise ValueError(f"Unsupported clustering method: {method}")
    unknown = set(kwargs) - _ALLOWED_KW
    if unknown:
        raise TypeError(f"Unsupported clustering keyword arguments: {sorted(unknown)}")
    rationale = None
    if Y.dtype not in (np.float32, np.float64):
        Y = Y.astype(np.float64)
    if n_states == "auto":
        n_states, rationale = _auto_select_n_states(Y, random_state, sample_size=silhouette_sample_size,
                                                    override_n_states=auto_n_states_override, kwargs=kwargs)
    k = int(n_states)
    if k < 1:
        raise ValueError("n_states must be >= 1")
    if Y.shape[0] < k:
        raise ValueError(f"Cannot create {k} clusters from {Y.shape[0]} samples")
    eng = get_engine()
    pipe = MSMPipeline(eng)
    yd = eng.to_device(np.ascontiguousarray(Y))
    seed = 0 if random_state is None else int(random_state)
    n_init = int(kwargs.get("n_init", 1))
    max_iter = int(kwargs.get("max_iter", 100))
    tol = float(kwargs.get("tolerance", 1e-5))
    best = None
    for r in range(max(1, n_init)):  # restarts keep the lowest inertia (clustering.py:584-629)
        labels, centers, inertia = _fit_once(pipe, yd, k, seed + r, max_iter, tol, kwargs.get("init_centers"))
        if best is None or inertia < best[2]:
            best = (labels, centers, inertia)
    raw = best[0].to_host()
    uniq, dense = np.unique(raw, return_inverse=True)
    n_unique = int(uniq.size)
    if n_unique < k:
        logger.warning("Clustering produced %d unique microstates (requested %d)", n_unique, k)
    Yf = np.asarray(Y, dtype=float)
    sums = np.zeros((n_unique, Y.shape[1]))
    np.add.at(sums, dense, Yf)
    centers = sums / np.bincount(dense, minlength=n_unique)[:, None]
    return ClusteringResult(labels=dense.astype(int), n_states=n_unique, rationale=rationale, centers=centers)
