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

__all__ = ["ClusteringResult", "cluster_microstates"]

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
    if n_states == "auto":
        if auto_n_states_override is None:
            raise NotImplementedError("n_states='auto' (silhouette scan) is outside the accelerated path; "
                                      "pass auto_n_states_override")
        n_states = int(auto_n_states_override)
        rationale = f"auto-override={n_states}"
    k = int(n_states)
    if k < 1:
        raise ValueError("n_states must be >= 1")
    if Y.shape[0] < k:
        raise ValueError(f"Cannot create {k} clusters from {Y.shape[0]} samples")
    if Y.dtype not in (np.float32, np.float64):
        Y = Y.astype(np.float64)
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
