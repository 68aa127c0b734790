"""MSM discretisation: mirror of pmarlo.analysis.discretize
(S/analysis/discretize.py:21-43 result, :406-514 _KMeansDiscretizer, :609-682 counting,
:901-1120 discretize_dataset).  Whitening statistics, k-means, assignment, counting and
row-normalisation run on the GPU; the split / segment / schema bookkeeping (a few Python
objects per split) stays on the host."""

from __future__ import annotations

import logging
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, List, Mapping, MutableMapping, Sequence

import numpy as np

from ..device import get_engine
from ..pipeline import MSMPipeline
from .counting import expected_pairs
from .validation import validate_features

logger = logging.getLogger("pmarlo")
DatasetLike = Mapping[str, Any] | MutableMapping[str, Any]

__all__ = ["MSMDiscretizationResult", "NoAssignmentsError", "discretize_dataset", "KMeansDiscretizer", "GridDiscretizer"]


@dataclass(slots=True)
class MSMDiscretizationResult:
    assignments: Dict[str, np.ndarray]
    centers: np.ndarray | None
    counts: np.ndarray
    transition_matrix: np.ndarray
    lag_time: int
    diag_mass: float
    cluster_mode: str
    assignment_masks: Dict[str, np.ndarray] = field(default_factory=dict)
    segment_lengths: Dict[str, List[int]] = field(default_factory=dict)
    segment_strides: Dict[str, List[int]] = field(default_factory=dict)
    counted_pairs: Dict[str, int] = field(default_factory=dict)
    expected_pairs: Dict[str, int] = field(default_factory=dict)
    feature_schema: Dict[str, Any] = field(default_factory=dict)
    fingerprint: Dict[str, Any] = field(default_factory=dict)
    feature_stats: Dict[str, Any] = field(default_factory=dict)
    counts_before_prune: np.ndarray | None = None
    state_counts_before_prune: np.ndarray | None = None
    state_counts: np.ndarray | None = None
    pruned_state_indices: np.ndarray | None = None


def _device_ready(X: np.ndarray) -> np.ndarray:
    """C-contiguous float32 / float64 as given (no re-typing pass over N x d on the host); anything else -> float64."""
    X = np.asarray(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    return np.ascontiguousarray(X)


class NoAssignmentsError(ValueError):
    """Defined by the reference but never raised there (it raises ValueError, :797)."""


def _looks_like_split(value: Any) -> bool:
    if isinstance(value, (Mapping, MutableMapping)):
        cand = value.get("X")
        if cand is None:
            return False
        arr = np.asarray(cand)
    elif hasattr(value, "X"):
        arr = np.asarray(getattr(value, "X"))
    else:
        arr = np.asarray(value)
    if arr.ndim != 2 or arr.shape[0] == 0 or arr.shape[1] == 0:
        return False
    return bool(np.isfinite(arr).all())


def _normalise_splits(dataset: DatasetLike) -> Dict[str, Any]:
    splits: Dict[str, Any] = {}
    maybe = dataset.get("splits") if isinstance(dataset, Mapping) else None
    if isinstance(maybe, Mapping):
        for name, value in maybe.items():
            if _looks_like_split(value):
                splits[str(name)] = value
    if not splits:
        for name, value in dataset.items():
            if not str(name).startswith("__") and _looks_like_split(value):
                splits[str(name)] = value
    if not splits and _looks_like_split(dataset):
        splits["all"] = dataset
    if not splits:
        raise ValueError("No continuous CV splits found in dataset")
    return splits


def _coerce_array(obj: Any) -> np.ndarray:
    arr = obj.get("X") if isinstance(obj, (Mapping, MutableMapping)) else getattr(obj, "X", obj)
    array = np.array(arr, dtype=np.float64, copy=False)
    if array.ndim != 2:
        raise ValueError(f"Expected 2D array, got shape {array.shape}")
    if array.shape[0] == 0:
        raise ValueError("Split is empty")
    return np.ascontiguousarray(array)


def _names(raw: Any) -> list[str]:
    if raw is None:
        return []
    if isinstance(raw, Mapping):
        raw = raw.get("names")
    if raw is None:
        return []
    if isinstance(raw, (str, bytes)):
        return [str(raw)]
    return [str(v) for v in raw if v is not None]


def _extract_feature_schema(split: Any, n_features: int) -> Dict[str, Any]:
    names: list[str] = []
    if isinstance(split, (Mapping, MutableMapping)):
        schema = split.get("feature_schema")
        if isinstance(schema, Mapping):
            names = _names(schema)
        if not names:
            cand = split.get("cv_names")
            names = _names(cand if cand is not None else split.get("feature_names"))
    if not names:
        names = [f"feature_{i}" for i in range(n_features)]
    return {"names": names, "n_features": int(n_features)}


def _validate_feature_schema(reference: Mapping[str, Any], candidate: Mapping[str, Any], *, split_name: str) -> None:
    exp_n, act_n = int(reference.get("n_features", 0)), int(candidate.get("n_features", 0))
    diffs: list[str] = []
    if act_n != exp_n:
        diffs.append(f"n_features mismatch: expected {exp_n}, got {act_n}")
    else:
        exp, act = list(reference.get("names") or []), list(candidate.get("names") or [])
        if exp and exp != act:
            diffs.append(f"names differ: expected {exp}, got {act}")
    if diffs:
        # the reference builds ValueError(..., differences=...) which itself raises TypeError (:224-229);
        # a plain ValueError carries the same information.
        raise ValueError(f"Feature schema mismatch for split '{split_name}': " + "; ".join(diffs))


def _segments_from_split_metadata(split: Mapping[str, Any]) -> tuple[list[int], list[int]]:
    lengths: list[int] = []
    strides: list[int] = []

    def add(length, stride):
        try:
            L = int(length)
        except Exception:
            return
        if L <= 0:
            return
        try:
            s = 1 if stride is None else int(stride)
        except Exception:
            s = 1
        lengths.append(L)
        strides.append(s if s > 0 else 1)

    meta = split.get("segments") or split.get("__segments__")
    if isinstance(meta, Iterable):
        for entry in meta:
            if isinstance(entry, Mapping):
                L = entry.get("length")
                if L is None and entry.get("start") is not None and entry.get("stop") is not None:
                    try:
                        L = int(entry["stop"]) - int(entry["start"])
                    except Exception:
                        L = None
                s = entry.get("stride") or entry.get("effective_frame_stride")
            else:
                L, s = entry, None
            if L is not None:
                add(L, s)
    if not lengths and isinstance(split.get("segment_lengths"), Iterable):
        for v in split["segment_lengths"]:
            add(v, None)
    return lengths, strides


def _truncate_segments(lengths, strides, total):
    consumed, out_l, out_s = 0, [], []
    for L, s in zip(lengths, strides):
        if consumed >= total:
            break
        v = min(int(L), total - consumed)
        if v <= 0:
            continue
        out_l.append(v)
        out_s.append(max(1, int(s)))
        consumed += v
    if not out_l and total > 0:
        return [total], [1]
    return out_l, out_s


def _lengths_to_segments(lengths: Sequence[int], total: int) -> list[tuple[int, int]]:
    segs, off = [], 0
    for v in lengths:
        L = int(v)
        if L <= 0:
            continue
        stop = min(total, off + L)
        if stop > off:
            segs.append((off, stop))
        off = stop
        if off >= total:
            break
    return segs or [(0, total)]


def _coerce_weights(weights: Any, n_frames: int, split_name: str) -> np.ndarray | None:
    if weights is None:
        return None
    cand = weights.get(split_name) if isinstance(weights, Mapping) else weights
    if cand is None:
        return None
    arr = np.asarray(cand, dtype=np.float64).reshape(-1)
    if arr.shape[0] != n_frames:
        raise ValueError(f"Frame weights for split '{split_name}' have length {arr.shape[0]}, expected {n_frames}")
    return arr


class KMeansDiscretizer:
    """_KMeansDiscretizer: whitening (mean, std ddof=1, std_safe) + k-means; ``fit`` runs the
    engine's Lloyd, or adopts ``centers`` (whitened space) when given (parity mode).

    The reference fits sklearn ``KMeans(n_init=10)`` (k-means++ starts) below 5e6 elements and ``MiniBatchKMeans``
    above (S/analysis/discretize.py:402-403, 458-469).  Here: below that size ``n_init`` restarts of full-batch Lloyd
    from k-means++ seeds on the device, lowest inertia kept; above it one full-batch Lloyd from the seeded stratified
    draw (where the reference trades quality for time, and so does the seeding: see cluster_microstates)."""

    _KMEANS_MAX_ELEMENTS = 5_000_000

    def __init__(self, n_states: int, *, random_state: int | None = None, apply_whitening: bool = True,
                 centers: np.ndarray | None = None, max_iter: int = 100, n_init: int = 3) -> None:
        self.n_states = int(n_states)
        self.n_init = max(1, int(n_init))
        self.random_state = random_state
        self.apply_whitening = bool(apply_whitening)
        self.max_iter = int(max_iter)
        self._given_centers = None if centers is None else np.ascontiguousarray(centers, np.float64)
        self.feature_schema: Dict[str, Any] | None = None
        self.scaler_mean_: np.ndarray | None = None
        self.scaler_std_: np.ndarray | None = None
        self._centers_d = None
        self._mean_d = None
        self._std_d = None
        self._eng = get_engine()

    def fit(self, X: np.ndarray, feature_schema: Mapping[str, Any] | None = None, *, device_array=None) -> None:
        """``device_array``: X already on the device (discretize_dataset uploads every split once)."""
        eng = self._eng
        n, d = X.shape
        schema = dict(feature_schema or {})
        if int(schema.get("n_features", d)) != d:
            raise ValueError(f"Feature schema reports {schema.get('n_features')} features, but training data has {d}")
        names = list(schema.get("names") or [])
        if names and len(names) != d:
            raise ValueError(f"Feature schema names length {len(names)} does not match n_features {d}")
        schema["names"] = [str(v) for v in names] if names else [f"feature_{i}" for i in range(d)]
        schema["n_features"] = d
        self.feature_schema = schema
        # float32 / float64 go up as they are: the kernels read both
        xd = device_array if device_array is not None else eng.to_device(_device_ready(X))
        whiten = None
        if self.apply_whitening:
            mean, std, _ = eng.column_moments(xd, ddof=1)
            self.scaler_mean_, self.scaler_std_ = mean.to_host(), std.to_host()
            std_safe = np.where(self.scaler_std_ > 1e-10, self.scaler_std_, 1.0)
            self._mean_d, self._std_d = mean, eng.to_device(std_safe)
            whiten = (self._mean_d, self._std_d)
        if self._given_centers is not None:
            if self._given_centers.shape != (self.n_states, d):
                raise ValueError("given centres have the wrong shape")
            self._centers_d = eng.to_device(self._given_centers)
        else:
            if n < self.n_states:
                raise ValueError(f"n_samples={n} should be >= n_clusters={self.n_states}.")
            pipe = MSMPipeline(eng)
            seed = int(self.random_state or 0)
            if n * d >= self._KMEANS_MAX_ELEMENTS:
                _, self._centers_d, _ = pipe.cluster(xd, self.n_states, seed=seed, max_iter=self.max_iter, tol=1e-4,
                                                     whiten=whiten)
            else:
                best = None
                md = eng.empty((n,), np.float64)
                mean_d, std_d = whiten if whiten is not None else (None, None)
                for r in range(self.n_init):
                    c0 = eng.kmeans_init_plusplus(xd, self.n_states, seed=seed + 7919 * r, mean=mean_d, std=std_d)
                    _, cen, _ = pipe.cluster(xd, self.n_states, seed=seed, max_iter=self.max_iter, tol=1e-4, whiten=whiten,
                                             centers=c0, mindist=md)
                    inertia = float(eng.sum_f64(md).to_host()[0])
                    if best is None or inertia < best[0]:
                        best = (inertia, cen)
                self._centers_d = best[1]

    def transform(self, X: np.ndarray, feature_schema: Mapping[str, Any] | None = None, *,
                  split_name: str | None = None, device_array=None, return_device: bool = False):
        if self._centers_d is None:
            raise RuntimeError("Discretizer has not been fitted")
        if feature_schema is not None and self.feature_schema is not None:
            _validate_feature_schema(self.feature_schema, feature_schema, split_name=split_name or "split")
        eng = self._eng
        xd = device_array if device_array is not None else eng.to_device(_device_ready(X))
        labels = eng.kmeans_assign(xd, self._centers_d, mean=self._mean_d, std=self._std_d)
        host = labels.to_host().astype(np.int32, copy=False)
        return (host, labels) if return_device else host

    @property
    def centers(self) -> np.ndarray | None:
        return None if self._centers_d is None else self._centers_d.to_host()

    @property
    def scaler_params(self) -> Dict[str, Any]:
        if not self.apply_whitening or self.scaler_mean_ is None:
            return {}
        return {"mean": self.scaler_mean_.tolist(), "std": self.scaler_std_.tolist(), "enabled": True}


class GridDiscretizer:
    """_GridDiscretizer (S/analysis/discretize.py:517-593): ``round(target ** (1/F))`` equal-width bins per
    feature between the training minimum and maximum; a frame's state is the order of first appearance
    of its bin combination (training frames first, then every transformed split in call order)."""

    _MAX_CELLS = 1 << 24

    def __init__(self, *, target_states: int) -> None:
        self.target_states = max(int(target_states), 1)
        self.edges: list[np.ndarray] = []
        self.feature_schema: Dict[str, Any] | None = None
        self._cell_map: np.ndarray | None = None     # cell -> state, -1 = not seen yet
        self._n_states = 0
        self._bins = 0
        self._eng = get_engine()

    def _assign_new_states(self, flat_d, n_cells: int) -> None:
        first = self._eng.first_occurrence(flat_d, n_cells)
        unseen = np.nonzero((first >= 0) & (self._cell_map < 0))[0]
        for cell in unseen[np.argsort(first[unseen], kind="stable")]:
            self._cell_map[cell] = self._n_states
            self._n_states += 1

    def fit(self, X: np.ndarray, feature_schema: Mapping[str, Any] | None = None) -> None:
        n, F = X.shape
        schema = dict(feature_schema or {})
        if int(schema.get("n_features", F)) != F:
            raise ValueError(f"Feature schema reports {schema.get('n_features')} features, but training data has {F}")
        names = list(schema.get("names") or [])
        schema["names"] = [str(v) for v in names] if names else [f"feature_{i}" for i in range(F)]
        schema["n_features"] = F
        self.feature_schema = schema
        bins = max(int(round(self.target_states ** (1.0 / F))), 1)
        if float(bins) ** F > self._MAX_CELLS:
            raise NotImplementedError(f"grid of {bins}^{F} cells is too large for the device table")
        eng = self._eng
        xd = eng.to_device(np.ascontiguousarray(X, np.float64))
        self.edges = []
        for col in range(F):
            st = eng.weighted_stats(xd, col)          # [., ., ., ., min, max]; NaN / inf propagate into the sum
            lo, hi = float(st[4]), float(st[5])
            if not np.isfinite(st[2]) or not np.isfinite(lo) or not np.isfinite(hi):
                raise ValueError("Non-finite values encountered while building grid")
            if lo == hi:
                lo, hi = lo - 0.5, hi + 0.5
            self.edges.append(np.linspace(lo, hi, bins + 1, dtype=np.float64))
        self._bins = bins
        n_cells = bins ** F
        self._cell_map = np.full(n_cells, -1, dtype=np.int32)
        self._n_states = 0
        self._assign_new_states(eng.grid_cells(xd, np.stack(self.edges)), n_cells)

    def transform(self, X: np.ndarray, feature_schema: Mapping[str, Any] | None = None, *,
                  split_name: str | None = None) -> np.ndarray:
        if not self.edges:
            raise RuntimeError("Discretizer has not been fitted")
        if feature_schema is not None and self.feature_schema is not None:
            _validate_feature_schema(self.feature_schema, feature_schema, split_name=split_name or "split")
        eng = self._eng
        flat = eng.grid_cells(eng.to_device(np.ascontiguousarray(X, np.float64)), np.stack(self.edges))
        self._assign_new_states(flat, self._cell_map.size)      # combinations not seen before get new states
        return eng.relabel(flat, self._cell_map).to_host()

    @property
    def centers(self) -> np.ndarray | None:
        if not self.edges:
            return None
        mesh = np.meshgrid(*[(e[:-1] + e[1:]) / 2.0 for e in self.edges], indexing="ij")
        return np.stack([m.ravel() for m in mesh], axis=1)

    @property
    def scaler_params(self) -> Dict[str, Any]:
        return {}


def _device_counts(labels: np.ndarray, n_states: int, lag: int, weights, segments, labels_device=None):
    eng = get_engine()
    pipe = MSMPipeline(eng)
    ld = labels_device if labels_device is not None else eng.to_device(np.ascontiguousarray(labels, np.int32))
    wd = eng.to_device(weights) if weights is not None else None
    counts, pairs = pipe.count(ld, n_states, lag, segments=segments, weights=wd)
    visits = eng.state_counts(ld, n_states).to_host()
    return counts, int(pairs.to_host()[0]), visits, ld


def discretize_dataset(dataset: DatasetLike, *, cluster_mode: str = "kmeans", n_microstates: int = 150,
                       lag_time: int = 1, frame_weights=None, min_out_count: int = 0,
                       random_state: int | None = None, apply_whitening: bool = True,
                       centers: np.ndarray | None = None) -> MSMDiscretizationResult:
    """Discretise continuous CVs into microstates and build MSM statistics.

    ``centers`` (whitened space, as MSMDiscretizationResult.centers) is an extension: it skips the
    fit so that results are reproducible against a reference fit (parity mode)."""
    if lag_time < 1:
        raise ValueError("lag_time must be >= 1")
    if cluster_mode not in ("kmeans", "grid"):
        raise ValueError("cluster_mode must be 'kmeans' or 'grid'")
    splits = _normalise_splits(dataset)
    train_key = "train" if "train" in splits else next(iter(splits))
    train_data = _coerce_array(splits[train_key])
    feature_schema = _extract_feature_schema(splits[train_key], train_data.shape[1])
    # every split goes to the device ONCE: validation, fit, assignment and counting share the copy
    kmeans = cluster_mode != "grid"
    on_device: Dict[str, Any] = {}

    def device_copy(name: str, X: np.ndarray):
        if name not in on_device:
            on_device[name] = get_engine().to_device(_device_ready(X))
        return on_device[name]

    stats_by_split: Dict[str, Dict[str, Any]] = {
        train_key: validate_features(train_data, feature_schema["names"], device_array=device_copy(train_key, train_data))}
    if not kmeans:
        disc = GridDiscretizer(target_states=n_microstates)
        disc.fit(train_data, feature_schema)
    else:
        disc = KMeansDiscretizer(n_microstates, random_state=random_state, apply_whitening=apply_whitening,
                                 centers=centers)
        disc.fit(train_data, feature_schema, device_array=on_device[train_key])
    feature_schema = disc.feature_schema or feature_schema
    stats_by_split[train_key]["feature_names"] = list(feature_schema["names"])
    stats_by_split[train_key]["n_features"] = int(feature_schema["n_features"])

    seg_len: Dict[str, List[int]] = {}
    seg_str: Dict[str, List[int]] = {}
    assignments: Dict[str, np.ndarray] = {}
    labels_on_device: Dict[str, Any] = {}
    masks: Dict[str, np.ndarray] = {}
    max_state = -1
    for name, split in splits.items():
        X = train_data if name == train_key else _coerce_array(split)
        schema = _extract_feature_schema(split, X.shape[1])
        _validate_feature_schema(feature_schema, schema, split_name=name)
        st = stats_by_split.get(name)
        if st is None:
            st = validate_features(X, schema["names"], device_array=device_copy(name, X))
            stats_by_split[name] = st
        st["feature_names"] = list(schema["names"])
        st["n_features"] = int(schema["n_features"])
        lengths, strides = ([], [])
        if isinstance(split, Mapping):
            lengths, strides = _segments_from_split_metadata(split)
        lengths, strides = _truncate_segments(lengths, strides, X.shape[0])
        st["segment_lengths"] = list(lengths)
        st["expected_pairs"] = expected_pairs(lengths, lag_time, strides if strides else 1)
        st["segment_strides"] = list(strides)
        if kmeans:
            labels, labels_on_device[name] = disc.transform(X, feature_schema=schema, split_name=name,
                                                            device_array=device_copy(name, X), return_device=True)
            on_device.pop(name, None)      # the coordinates of this split are not needed again
        else:
            labels = disc.transform(X, feature_schema=schema, split_name=name)
        valid = labels >= 0
        if not valid.any():
            raise ValueError(f"No valid assignments found for split '{name}'")
        assignments[name], masks[name] = labels, valid.astype(bool)
        seg_len[name], seg_str[name] = list(lengths), list(strides)
        if labels.size:
            max_state = max(max_state, int(labels.max()))
    n_states = max_state + 1 if max_state >= 0 else 0

    train_labels = assignments[train_key]
    if not masks[train_key].all():
        raise ValueError(f"No valid assignments found for split '{train_key}'")
    weights = _coerce_weights(frame_weights, train_labels.size, train_key)
    train_lengths = seg_len.get(train_key) or [train_labels.size]
    train_strides = seg_str.get(train_key) or []
    train_segments = _lengths_to_segments(train_lengths, train_labels.size)

    counts_d, counted, visits, _ = _device_counts(train_labels, n_states, lag_time, weights, train_segments,
                                                  labels_device=labels_on_device.get(train_key))
    counts = counts_d.to_host().astype(np.float64)
    counts_before = counts.copy()
    counted_before = counted
    exp_pairs = expected_pairs(train_lengths, lag_time, train_strides if train_strides else 1)
    if weights is not None:
        state_counts_before = np.bincount(train_labels, weights=weights, minlength=n_states).astype(np.float64)
    else:
        state_counts_before = visits.astype(np.float64)

    row_sums_before = counts_before.sum(axis=1)
    zero_rows_before = int(np.count_nonzero(row_sums_before == 0))
    min_out = max(0, int(min_out_count))
    pruned = None
    zero_rows_after = zero_rows_before
    if zero_rows_before > 0:  # _prune_zero_rows_if_needed (:825-898)
        prune = row_sums_before == 0
        if min_out > 0:
            prune |= row_sums_before < float(min_out)
        keep = ~prune
        if not keep.any():
            raise RuntimeError(f"Pruning removed all microstates (zero_rows={zero_rows_before}, min_out_count={min_out})")
        pruned = np.where(prune)[0].astype(np.int32)
        mapping = np.full(n_states, -1, dtype=np.int32)
        mapping[keep] = np.arange(int(keep.sum()), dtype=np.int32)
        for nm, lab in list(assignments.items()):
            rem = np.full_like(lab, -1)
            ok = (lab >= 0) & (lab < mapping.size)
            rem[ok] = mapping[lab[ok]]
            assignments[nm] = rem.astype(np.int32)
            masks[nm] = masks[nm] & (rem >= 0)
        train_labels = assignments[train_key]
        n_states = int(keep.sum())
        counts_d, counted, visits, _ = _device_counts(train_labels, n_states, lag_time, weights, train_segments)
        counts = counts_d.to_host().astype(np.float64)
        zero_rows_after = int(np.count_nonzero(counts.sum(axis=1) == 0))
        if zero_rows_after > 0:
            raise RuntimeError(f"Pruning left {zero_rows_after} zero-row microstates (min_out_count={min_out})")
    if weights is not None:
        okl = train_labels >= 0
        state_counts_final = np.bincount(train_labels[okl], weights=weights[okl], minlength=n_states).astype(np.float64)
    else:
        state_counts_final = visits.astype(np.float64)

    if exp_pairs > 0 and counted == 0:
        raise ValueError(f"No transition pairs counted for split '{train_key}' despite expected {exp_pairs} pairs")
    st = stats_by_split[train_key]
    st.update(expected_pairs=int(exp_pairs), counted_pairs_before_prune=int(counted_before), counted_pairs=int(counted),
              zero_rows_before_prune=int(zero_rows_before), zero_rows_after_prune=int(zero_rows_after))
    if pruned is not None and pruned.size:
        st["pruned_state_indices"] = pruned.astype(int).tolist()
        st["prune_min_out_count"] = int(min_out)

    eng = get_engine()
    tm = eng.transition_matrix(eng.to_device(counts), mode=0)
    transition = tm["T"].to_host()
    diag_mass = float(tm["diag_mass"].to_host()[0]) if n_states else float("nan")
    if np.isfinite(diag_mass) and diag_mass > 0.95:
        logger.warning("MSM diagonal mass high (%.3f)", diag_mass)

    fingerprint = {
        "mode": str(cluster_mode), "n_states": int(max(n_states, 0)),
        "seed": None if random_state is None else int(random_state),
        "feature_schema": {"names": list(feature_schema["names"]), "n_features": int(feature_schema["n_features"])},
        "expected_pairs": int(exp_pairs), "counted_pairs": int(counted),
        "segment_lengths": {train_key: train_lengths}, "segment_strides": {train_key: seg_str.get(train_key, [])},
        "zero_rows_before_prune": int(zero_rows_before), "zero_rows_after_prune": int(zero_rows_after),
        "pruned_state_count": int(pruned.size) if pruned is not None else 0, "min_out_count": int(min_out),
        "scaler": disc.scaler_params,
    }
    return MSMDiscretizationResult(
        assignments=assignments, assignment_masks=masks, segment_lengths=seg_len, segment_strides=seg_str,
        counted_pairs={train_key: int(counted)}, expected_pairs={train_key: int(exp_pairs)}, centers=disc.centers,
        counts=counts, transition_matrix=transition, lag_time=lag_time, diag_mass=diag_mass, cluster_mode=cluster_mode,
        feature_schema=feature_schema, fingerprint=fingerprint, feature_stats=stats_by_split,
        counts_before_prune=counts_before, state_counts_before_prune=state_counts_before,
        state_counts=state_counts_final, pruned_state_indices=pruned)
