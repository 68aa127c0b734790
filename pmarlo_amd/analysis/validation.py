"""validate_features: mirror of pmarlo.analysis.validation (S/analysis/validation.py:89-172).
Column statistics come from the device (msm_column_moments, msm_column_minmax)."""

from __future__ import annotations

import json
from typing import Any, Dict, List, Sequence

import numpy as np

from ..device import get_engine

__all__ = ["ValidationError", "validate_features"]

NUMERIC_MIN_POSITIVE = 1e-12


def _json_default(obj: Any) -> Any:
    if isinstance(obj, np.generic):
        return obj.item()
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    return obj


class ValidationError(RuntimeError):
    def __init__(self, message: str, *, code: str, stats: Dict[str, Any]) -> None:
        self.code = str(code)
        self.stats = dict(stats)
        summary = json.dumps(self.stats, sort_keys=True, default=_json_default)
        super().__init__(f"{message} [code={self.code}] stats={summary}")


def validate_features(X, feature_names: Sequence[str] | None, *, device_array=None) -> Dict[str, Any]:
    """Statistics and checks of a CV matrix.  The passes over the matrix run on the device (column moments; minima,
    maxima, non-finite entries and fully finite rows in one more pass): ``device_array`` is the matrix already on the
    device (discretize_dataset uploads every split once and shares it with the discretizer).  Only a matrix WITH
    non-finite entries -- the error path -- is walked on the host, column by column, for the per-column statistics of
    the finite subsets the error message carries."""
    array = np.asarray(X)
    if array.dtype not in (np.float32, np.float64):
        array = array.astype(np.float64)
    if array.ndim != 2:
        raise ValueError(f"Expected 2D feature matrix, got shape {array.shape}")
    n_rows, n_features = array.shape
    names: List[str] = ([f"feature_{i}" for i in range(n_features)] if feature_names is None
                        else [str(v) for v in feature_names][:n_features])
    names += [f"feature_{i}" for i in range(len(names), n_features)]
    if n_rows > 0 and n_features > 0:
        eng = get_engine()
        xd = device_array if device_array is not None else eng.to_device(np.ascontiguousarray(array))
        mn, mx, cnt = eng.column_minmax(xd)
        non_finite, finite_rows = (int(v) for v in cnt.to_host())
    else:
        non_finite, finite_rows = 0, int(n_rows)
    if non_finite == 0 and n_rows > 0 and n_features > 0:
        mean, std, _ = eng.column_moments(xd, ddof=0)
        means, stds = mean.to_host().tolist(), std.to_host().tolist()
        mins, maxs = mn.to_host().tolist(), mx.to_host().tolist()
    else:  # degenerate input: per-column finite subsets (error path)
        array = array.astype(np.float64, copy=False)
        finite = np.isfinite(array)
        means, stds, mins, maxs = [], [], [], []
        for j in range(n_features):
            col = array[:, j][finite[:, j]]
            if col.size == 0:
                means.append(float("nan")); stds.append(float("nan")); mins.append(float("nan")); maxs.append(float("nan"))
            else:
                means.append(float(col.mean())); stds.append(float(col.std())); mins.append(float(col.min())); maxs.append(float(col.max()))
    stats: Dict[str, Any] = {"feature_names": names, "n_rows": int(n_rows), "n_features": int(n_features),
                             "finite_rows": finite_rows, "non_finite_entries": non_finite, "means": means,
                             "stds": stds, "mins": mins, "maxs": maxs}
    if finite_rows == 0:
        raise ValidationError("No rows with fully finite CV values detected", code="cv_no_finite_rows", stats=stats)
    if non_finite > 0:
        raise ValidationError("CV matrix contains non-finite values", code="cv_non_finite", stats=stats)
    bad = [nm for nm, sd in zip(names, stds) if not np.isfinite(sd) or sd <= NUMERIC_MIN_POSITIVE]
    if bad:
        extra = {"problematic_features": bad}
        extra.update(stats)
        raise ValidationError("Detected CV columns with zero or invalid standard deviation", code="cv_zero_std",
                              stats=extra)
    return stats
