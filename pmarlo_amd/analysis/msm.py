"""prepare_msm_discretization: mirror of pmarlo.analysis.msm (S/analysis/msm.py:53-106).
Like the reference, ``apply_whitening`` is not forwarded to discretize_dataset (:74-82)."""

from __future__ import annotations

from typing import Any, Mapping, MutableMapping

import numpy as np

from .discretize import MSMDiscretizationResult, discretize_dataset

__all__ = ["prepare_msm_discretization"]


def prepare_msm_discretization(dataset, *, cluster_mode: str = "kmeans", n_microstates: int = 150, lag_time: int = 1,
                               frame_weights: Mapping[str, Any] | Any | None = None, min_out_count: int = 0,
                               random_state: int | None = None, apply_whitening: bool = True,
                               centers: np.ndarray | None = None) -> MSMDiscretizationResult:
    if isinstance(dataset, (MutableMapping, dict)):
        art = dataset.get("__artifacts__")
        if apply_whitening and isinstance(art, Mapping) and isinstance(art.get("mlcv_deeptica"), Mapping):
            raise NotImplementedError("DeepTICA output whitening (ensure_msm_inputs_whitened) is outside the "
                                      "accelerated path; whiten the CVs before calling")
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
