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

# keyword arguments of the reference (clustering.py:236-262): what deeptime's estimators take
_COMMON_KWARGS = frozenset({"max_iter", "metric", "tolerance", "init_strategy", "n_jobs", "initial_centers"})
_ATTRIBUTE_KWARGS = frozenset({"fixed_seed", "progress"})
_MINIBATCH_ONLY_KWARGS = frozenset({"batch_size"})
_SUPPORTED_KWARGS = _COMMON_KWARGS | _ATTRIBUTE_KWARGS | _MINIBATCH_ONLY_KWARGS
# extensions of this engine: n_init is the reference's own restart count (popped before validation there too);
# init_centers is the round-1 spelling of initial_centers
_EXTENSION_KWARGS = frozenset({"n_init", "init_centers"})


@dataclass
class ClusteringResult:
    labels: np.ndarray
    n_states: int
    rationale: str | None = None
    centers: np.ndarray | None = None

    @property
    def output_shape(self) -> tuple[int, ...]:
        return (self.n_states,)


def _fit_once(pipe, yd, k, seed, max_iter, tol, init, init_strategy="kmeans++", minibatch=None):
    """One fit from one start: `init` (given centres), else k-means++ seeding or the seeded stratified draw; `minibatch`
    = batch size of the mini-batch estimator (None: full-batch Lloyd)."""
    eng = pipe.eng
    md = eng.empty((yd.shape[0],), np.float64)
    centers0 = eng.to_device(np.ascontiguousarray(init, np.float64)) if init is not None else None
    if centers0 is None and init_strategy == "kmeans++" and minibatch is None:
        centers0 = eng.kmeans_init_plusplus(yd, k, seed=seed)
    if minibatch is not None:
        centers0, _ = eng.kmeans_fit_minibatch(yd, k, seed=seed, batch_size=minibatch, max_iter=max_iter,
                                               init=init_strategy, centers=centers0)
        labels, centers, _ = pipe.cluster(yd, k, centers=centers0, mindist=md, fit=False)
        return labels, centers, float(eng.sum_f64(md).to_host()[0])
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
                                           float(kwargs.get("tolerance", 1e-5)), None,
                                           _init_strategy(kwargs, int(Ys.size)))
            if best is None or inertia < best[1]:
                best = (labels, inertia)
        lab = best[0].to_host()
        scores.append((k, silhouette_score(Ys64, lab) if np.unique(lab).size > 1 else -1.0))
    if not scores:
        raise ValueError("too few samples for the n_states='auto' scan (needs more than 4)")
    chosen, best = max(scores, key=lambda t: t[1])
    logger.info("Auto-selected %d states with silhouette score %.3f", chosen, best)
    return int(chosen), f"silhouette={best:.3f}{note}"


def _validate_clustering_kwargs(method: str, kwargs: dict) -> None:
    """clustering.py:246-262: unknown keywords and batch_size with method='kmeans' are TypeErrors."""
    unsupported = set(kwargs) - _SUPPORTED_KWARGS - _EXTENSION_KWARGS
    if unsupported:
        raise TypeError(f"Unsupported clustering parameters for deeptime backend: {sorted(unsupported)}")
    if method == "kmeans" and any(k in kwargs for k in _MINIBATCH_ONLY_KWARGS):
        raise TypeError("'batch_size' is only supported when method='minibatchkmeans'.")
    metric = kwargs.get("metric", "euclidean")
    if metric not in ("euclidean", None):
        raise NotImplementedError(f"metric={metric!r}: the device k-means is Euclidean (deeptime's default)")


def _init_strategy(kwargs: dict, n_elements: int = 0, threshold: int = 5_000_000) -> str:
    """deeptime's KMeans takes init_strategy 'kmeans++' (its default, and what the reference gets) or 'uniform'
    (k frames at random: here the seeded stratified draw along the time axis).  Without the keyword: k-means++ for
    inputs the reference clusters with full-batch KMeans, the stratified draw above ``minibatch_threshold`` where the
    reference itself gives up quality for time (k-means++ costs two launches and a pass over the data per centre:
    15 ms at 1 M x 10, k = 500, three times the whole fit)."""
    strategy = kwargs.get("init_strategy", "kmeans++" if n_elements <= threshold else "uniform")
    if strategy not in ("kmeans++", "uniform"):
        raise ValueError(f"init_strategy must be 'kmeans++' or 'uniform', got {strategy!r}")
    return strategy


def _resolve_seed(random_state: int | None, kwargs: dict) -> int:
    """clustering.py:289-304 (_resolve_fixed_seed): an integer fixed_seed is the seed, True is deeptime's fixed
    seed 42, False / no random_state mean "not fixed" (the engine is deterministic anyway: seed 0)."""
    if "fixed_seed" in kwargs:
        fixed = kwargs["fixed_seed"]
        if isinstance(fixed, bool):
            return 42 if fixed else 0
        if isinstance(fixed, (int, np.integer)):
            return int(fixed)
        raise TypeError(f"'fixed_seed' must be an integer or boolean, received type {type(fixed)!r}.")
    return 0 if random_state is None else int(random_state)


def _restart_seeds(random_state: int | None, n_init: int) -> list[int | None]:
    """The reference's seed list for n_init restarts (clustering.py:586-603)."""
    if n_init == 1:
        return [random_state]
    rng = np.random.default_rng(random_state)
    seeds: list[int | None] = [None if random_state is None else int(random_state)]
    existing = {s for s in seeds if isinstance(s, int)}
    while len(seeds) < n_init:
        cand = int(rng.integers(0, np.iinfo(np.int32).max))
        if cand in existing:
            continue
        seeds.append(cand)
        existing.add(cand)
    return seeds


def cluster_microstates(
    Y: np.ndarray,
    method: Literal["auto", "minibatchkmeans", "kmeans"] = "auto",
    n_states: int | Literal["auto"] = "auto",
    random_state: int | None = 42,
    minibatch_threshold: int = 5_000_000,
    *,
    silhouette_sample_size: int | None = None,
    auto_n_states_override: int | None = None,
    **kwargs: Any,
) -> ClusteringResult:
    """k-means microstates on the GPU; same signature, keyword set, errors and result as the reference
    (S/markov_state_model/clustering.py:395-665).

    ``method="kmeans"`` is full-batch Lloyd, ``method="minibatchkmeans"`` the mini-batch estimator
    (``batch_size``, default 100 as deeptime's; ``max_iter`` sweeps over the batches), ``method="auto"`` always
    full-batch: the reference switches to mini-batch above ``minibatch_threshold`` to bound CPU time, which is not
    a concern here (a full-batch pass over 1 M x 10 takes 0.1 ms and ends at a lower inertia).  The start is
    ``init_strategy``: "kmeans++" (deeptime's default: D^2 seeding on the device) or "uniform" (seeded stratified
    draw), or ``initial_centers``; ``n_jobs`` and ``progress`` have no meaning on the device and are ignored.
    Labels are densified and centres recomputed as member
    means as the reference does after its estimator returns (:364-392) -- on the device: member sums come out of
    one more accumulate pass in exact fixed point, the relabelling is a device gather."""
    Y = np.asarray(Y)
    if Y.ndim == 2 and Y.shape[0] == 0:
        logger.info("Empty dataset provided, returning empty clustering result")
        return ClusteringResult(labels=np.empty((0,), dtype=int), n_states=0)
    if Y.ndim != 2:
        raise ValueError(f"Input must be 2D array, got shape {Y.shape}")
    if Y.shape[1] == 0:
        raise ValueError("Input array must have at least one feature")
    kwargs = dict(kwargs)
    raw_n_init = kwargs.get("n_init")
    if raw_n_init is None:
        n_init = 1
    else:
        try:
            n_init = int(raw_n_init)
        except (TypeError, ValueError) as exc:
            raise TypeError("n_init must be provided as an integer when clustering with deeptime") from exc
        if n_init <= 0:
            raise ValueError("n_init must be a positive integer when clustering microstates")
    if n_init > 1 and "fixed_seed" in kwargs:
        raise ValueError("n_init cannot be combined with fixed_seed; provide only one mechanism "
                         "for controlling clustering initialisations.")
    _validate_clustering_kwargs(method, kwargs)
    if method not in ("auto", "minibatchkmeans", "kmeans"):
        raise ValueError(f"Unsupported clustering method: {method}")
    strategy = _init_strategy(kwargs, int(Y.shape[0] * Y.shape[1]), int(minibatch_threshold))
    for ignored in ("n_jobs", "progress"):
        if ignored in kwargs:
            logger.debug("cluster_microstates: %s=%r has no effect on the device estimator", ignored, kwargs[ignored])
    rationale = None
    if Y.dtype not in (np.float32, np.float64):
        Y = Y.astype(np.float64)          # float32 / float64 go to the device as they are (the kernels read both)
    requested = n_states
    if isinstance(n_states, str):
        if n_states != "auto":
            raise ValueError(f"n_states must be an integer or 'auto', got {n_states!r}")
        n_states, rationale = _auto_select_n_states(Y, random_state, sample_size=silhouette_sample_size,
                                                    override_n_states=auto_n_states_override, kwargs=kwargs)
    k = int(n_states)
    if k <= 0:
        raise ValueError(f"Number of microstates must be a positive integer; received {k}.")
    # "auto": the reference would pick mini-batch above the threshold; the device estimator stays full-batch (docstring)
    would = "minibatchkmeans" if int(Y.shape[0] * Y.shape[1]) > minibatch_threshold else "kmeans"
    chosen = "kmeans" if method == "auto" else method
    if method == "auto" and would == "minibatchkmeans":
        if "batch_size" in kwargs:
            chosen = "minibatchkmeans"      # the caller asked for batches and the reference would use them here
        else:
            logger.info("cluster_microstates: %d x %d exceeds minibatch_threshold; the device estimator runs full-batch "
                        "Lloyd all the same", Y.shape[0], Y.shape[1])
    if "batch_size" in kwargs and chosen != "minibatchkmeans":
        raise ValueError(f"batch_size was provided but the selected clustering method is '{chosen}'. "
                         "Specify method='minibatchkmeans' to use mini-batch parameters.")
    if Y.shape[0] < k:
        raise ValueError(f"Cannot create {k} clusters from {Y.shape[0]} samples")
    eng = get_engine()
    pipe = MSMPipeline(eng)
    yd = eng.to_device(np.ascontiguousarray(Y))
    max_iter = int(kwargs.get("max_iter", 5 if chosen == "minibatchkmeans" else 100))
    tol = float(kwargs.get("tolerance", 1e-5))
    init = kwargs.get("initial_centers", kwargs.get("init_centers"))
    if init is not None:
        init = np.asarray(init, dtype=np.float64)
        if init.shape != (k, Y.shape[1]):
            raise ValueError(f"initial_centers must have shape {(k, Y.shape[1])}, got {init.shape}")
    best = None
    for it, seed in enumerate(_restart_seeds(random_state, n_init)):  # restarts keep the lowest inertia (:584-629)
        s = _resolve_seed(seed, kwargs)
        labels, centers, inertia = _fit_once(pipe, yd, k, s, max_iter, tol, init, strategy,
                                             int(kwargs.get("batch_size", 100)) if chosen == "minibatchkmeans" else None)
        if best is None or inertia < best[2]:
            best = (labels, centers, inertia, seed, it)
    if n_init > 1:
        logger.info("Selected best clustering from %d initialisations (iteration=%d, seed=%s, inertia=%.6f)", n_init,
                    best[4], "None" if best[3] is None else int(best[3]), float(best[2]))
    labels_d, centers_d = best[0], best[1]
    # ---- densify + member means (_remap_labels_and_compute_inertia :364-392) on the device
    n, d = Y.shape
    _, st = eng.kmeans_fit_begin(yd, k, seed=0, n_total=n, tol2=0.0, centers=centers_d, init_centers=False)
    sums, counts = eng.zeros((k * d,), np.int64), eng.zeros((k,), np.int64)
    eng.kmeans_accumulate(yd, centers_d, st, sums, counts)
    cnt = counts.to_host()
    occupied = np.flatnonzero(cnt > 0)
    n_unique = int(occupied.size)
    if n_unique == 0:
        raise ValueError("Clustering produced zero unique microstates; verify input coverage and CV preprocessing.")
    inv_scale = float(st.to_host()[1])
    centers = sums.to_host().reshape(k, d)[occupied].astype(np.float64) * inv_scale / cnt[occupied, None]
    if n_unique != k:
        logger.warning("Clustering produced %d unique microstates, expected %d. Proceeding with the observed value; "
                       "inspect CV spread or adjust the requested microstate count.", n_unique, k)
        dense_map = np.full(k, -1, np.int32)
        dense_map[occupied] = np.arange(n_unique, dtype=np.int32)
        labels_d = eng.relabel(labels_d, dense_map)
    logger.info("Clustering completed: requested=%s, actual=%d%s", requested, n_unique,
                f" ({rationale})" if rationale else "")
    return ClusteringResult(labels=labels_d.to_host().astype(int), n_states=n_unique, rationale=rationale, centers=centers)
