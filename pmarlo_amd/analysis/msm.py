"""prepare_msm_discretization: mirror of pmarlo.analysis.msm (S/analysis/msm.py:53-106).
``apply_whitening`` whitens the top-level "X" from DeepTICA metadata when present (ensure_msm_inputs_whitened
:18-50); like the reference it is not forwarded to discretize_dataset (:74-82)."""

from __future__ import annotations

from typing import Any, Mapping, MutableMapping

import numpy as np

from .discretize import MSMDiscretizationResult, discretize_dataset

__all__ = ["prepare_msm_discretization", "ensure_msm_inputs_whitened"]


def ensure_msm_inputs_whitened(dataset) -> bool:
    """Whiten dataset["X"] with the DeepTICA output transform recorded under
    dataset["__artifacts__"]["mlcv_deeptica"], once (S/analysis/msm.py:18-50).  True when it was applied."""
    from .project_cv import apply_whitening_from_metadata

    if not isinstance(dataset, (MutableMapping, dict)):
        return False
    X = dataset.get("X")
    if X is None:
        return False
    artifacts = dataset.get("__artifacts__")
    summary = artifacts.get("mlcv_deeptica") if isinstance(artifacts, Mapping) else None
    if not isinstance(summary, Mapping):
        return False
    if summary.get("output_mean") is None or summary.get("output_transform") is None:
        return False
    whitened, applied = apply_whitening_from_metadata(np.asarray(X, dtype=np.float64), summary)
    if applied:
        dataset["X"] = whitened
    return applied


def prepare_msm_discretization(dataset, *, cluster_mode: str = "kmeans", n_microstates: int = 150, lag_time: int = 1,
                               frame_weights: Mapping[str, Any] | Any | None = None, min_out_count: int = 0,
                               random_state: int | None = None, apply_whitening: bool = True,
                               centers: np.ndarray | None = None) -> MSMDiscretizationResult:
    if isinstance(dataset, (MutableMapping, dict)):
        if apply_whitening:
            ensure_msm_inputs_whitened(dataset)
        if frame_weights is None and dataset.get("frame_weights") is not None:
            frame_weights = dataset.get("frame_weights")
    result = discretize_dataset(dataset, cluster_mode=cluster_mode, n_microstates=n_microstates, lag_time=lag_time,
                                frame_weights=frame_weights, min_out_count=min_out_count, random_state=random_state,
                                centers=centers)
    if isinstance(dataset, MutableMapping):
        artifacts = dataset.setdefault("__artifacts__", {})
        if isinstance(artifacts, MutableMapping):
            artifacts["feature_stats"] = result.feature_stats
            artifacts["state_assignments"] = {s: {"n_assigned": int(np.count_nonzero(m)), "total": int(m.size)}
                                              for s, m in result.assignment_masks.items()}
            artifacts["segment_lengths"] = result.segment_lengths
            artifacts["segment_strides"] = result.segment_strides
            artifacts["expected_pairs"] = result.expected_pairs
            artifacts["counted_pairs"] = result.counted_pairs
            if result.pruned_state_indices is not None:
                artifacts["pruned_state_indices"] = result.pruned_state_indices.tolist()
            if result.state_counts is not None:
                artifacts["state_counts_post_prune"] = result.state_counts.tolist()
            if result.state_counts_before_prune is not None:
                artifacts["state_counts_pre_prune"] = result.state_counts_before_prune.tolist()
    return result
