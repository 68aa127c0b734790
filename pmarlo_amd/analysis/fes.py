"""Frame-weighted free-energy surfaces: mirror of pmarlo.analysis.fes.compute_weighted_fes
(S/analysis/fes.py:411-453; :20-88 component selection, :91-114 weights, :142-173 bandwidth,
:176-238 KDE, :241-292 histogram + sparse-bin smoothing, :570-599 free energy) with the frame
passes on the GPU: column variances, weighted statistics, the 2-D histogram (LDS bins, integer
atomics) and the Gaussian KDE (fp64 matrix cores, frames as the contraction index).

``apply_whitening`` from DeepTICA metadata (ensure_fes_inputs_whitened, :295-380) runs through
analysis/project_cv.py (device projection / moment passes)."""

from __future__ import annotations

from typing import Any, Mapping, MutableMapping, Sequence

import numpy as np

from ..device import get_engine

__all__ = ["compute_weighted_fes", "select_highest_variance_components"]

BOLTZMANN_CONSTANT_KJ_PER_MOL = 0.00831446261815324  # S/constants.py:16


def _pick_components(variances: np.ndarray, n_components: int) -> list[int]:
    if n_components < 1:
        raise ValueError("Must select at least one component")
    if variances.size < n_components:
        raise ValueError(f"Coordinate array has {variances.size} dimensions, need {n_components}")
    idx = np.where(variances > 0)[0]
    if idx.size < n_components:
        raise ValueError("FES component selection requires at least "
                         f"{n_components} non-constant CV columns; found {idx.size}")
    return idx[np.argsort(variances[idx])[::-1]][:n_components].tolist()


def select_highest_variance_components(coords: np.ndarray, n_components: int = 2) -> tuple[np.ndarray, list[int]]:
    """Columns of highest population variance (non-constant ones only), in descending order."""
    coords = np.asarray(coords)
    if coords.ndim != 2:
        raise ValueError(f"Expected 2D coordinate array, got shape {coords.shape}")
    if coords.shape[1] < 1:
        raise ValueError("Coordinate array must have at least one dimension")
    eng = get_engine()
    _, std, _ = eng.column_moments(eng.to_device(np.ascontiguousarray(coords, np.float64)), ddof=0)
    sel = _pick_components(std.to_host() ** 2, n_components)
    return coords[:, sel], sel


def _select_split(dataset: Mapping[str, Any], split: str | None) -> tuple[str, Mapping[str, Any]]:
    splits = dataset.get("splits")
    if not isinstance(splits, Mapping) or not splits:
        raise ValueError("Dataset must define at least one split with CV data")
    if split is None:
        name = next(iter(splits))
        return str(name), splits[name]
    if split not in splits:
        raise KeyError(f"Split {split!r} not found in dataset")
    return str(split), splits[split]


def _bins_pair(bins) -> tuple[int, int]:
    if isinstance(bins, (int, np.integer)):
        return int(bins), int(bins)
    if isinstance(bins, Sequence) and len(bins) == 2 and all(isinstance(b, (int, np.integer)) for b in bins):
        return int(bins[0]), int(bins[1])
    raise TypeError("bins must be an int or a sequence of two ints (explicit edge arrays are not supported)")


def _outer_edges(lo: float, hi: float, n: int) -> np.ndarray:
    """np.histogram's automatic range: [min, max], widened by 0.5 on both sides when degenerate."""
    if not (np.isfinite(lo) and np.isfinite(hi)):
        raise ValueError(f"autodetected range of [{lo}, {hi}] is not finite")
    if lo == hi:
        lo, hi = lo - 0.5, hi + 0.5
    return np.linspace(lo, hi, n + 1, dtype=np.float64)


def _bandwidth(var: float, ess: float, selector) -> float:
    if isinstance(selector, (int, float)):
        if float(selector) <= 0:
            raise ValueError("Bandwidth must be positive")
        return float(selector)
    if var <= 0.0:
        raise ValueError("Coordinate variance must be positive to compute bandwidth")
    n_eff = max(ess, 1.0)
    name = str(selector).lower()
    if name == "scott":
        factor = n_eff ** (-1.0 / 6.0)
    elif name == "silverman":
        factor = (n_eff * 4.0 / 4.0) ** (-1.0 / 6.0)
    else:
        raise ValueError("Bandwidth must be 'scott', 'silverman', or a positive float")
    bw = float(np.sqrt(var) * factor)
    if not np.isfinite(bw) or bw <= 0.0:
        raise ValueError("Computed bandwidth must be finite and positive")
    return bw


def ensure_fes_inputs_whitened(dataset) -> bool:
    """Apply the DeepTICA output whitening recorded under dataset["__artifacts__"]["mlcv_deeptica"] to the
    top-level "X" and then to every split's "X" (S/analysis/fes.py:295-380).  False when there is nothing to
    apply (no artifacts, no metadata, no top-level X, or metadata without a transform)."""
    from .project_cv import apply_whitening_from_metadata

    if not isinstance(dataset, (MutableMapping, dict)):
        raise TypeError("Dataset must be a mutable mapping to apply whitening")
    artifacts = dataset.get("__artifacts__")
    if artifacts is None:
        return False
    if not isinstance(artifacts, Mapping):
        raise ValueError(f"Dataset __artifacts__ must be a Mapping, got {type(artifacts)}")
    summary = artifacts.get("mlcv_deeptica")
    if summary is None or "X" not in dataset:
        return False
    if dataset["X"] is None:
        raise ValueError("Dataset provides no coordinate array for whitening")
    if not isinstance(summary, (MutableMapping, dict)):
        raise ValueError(f"mlcv_deeptica metadata must be a dict, got {type(summary)}")
    if summary.get("output_mean") is None or summary.get("output_transform") is None:
        return False
    whitened, applied = apply_whitening_from_metadata(np.asarray(dataset["X"], dtype=np.float64), summary)
    dataset["X"] = whitened
    applied_any = bool(applied)
    if applied:
        splits = dataset.get("splits")
        if isinstance(splits, Mapping):
            for name, data in splits.items():
                if not isinstance(data, MutableMapping) or "X" not in data:
                    continue
                if data["X"] is None:
                    raise ValueError(f"Split '{name}' provides no coordinate array for whitening")
                summary["output_transform_applied"] = False           # every split takes the transform once
                out, did = apply_whitening_from_metadata(np.asarray(data["X"], dtype=np.float64), summary)
                data["X"] = out
                applied_any = applied_any or bool(did)
            summary["output_transform_applied"] = True
    return applied_any


def compute_weighted_fes(dataset, *, split: str | None = None, weights=None, bins=64, temperature_K: float = 300.0,
                         method: str = "kde", bandwidth: str | float = "scott", min_count_per_bin: int = 1,
                         apply_whitening: bool = True) -> dict[str, Any]:
    """{"histogram", "xedges", "yedges", "free_energy", "metadata"} on the two highest-variance CVs
    of the split; ``method`` "kde" (Gaussian product kernel on the bin centres) or "grid"."""
    if not isinstance(dataset, (MutableMapping, dict)):
        raise ValueError("Dataset must be a mapping with 'splits'")
    if apply_whitening:
        ensure_fes_inputs_whitened(dataset)
    split_name, split_data = _select_split(dataset, split)
    if "X" not in split_data:
        raise KeyError("Split is missing 'X'")
    X = np.asarray(split_data["X"], dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    if X.ndim != 2 or X.shape[0] == 0:
        raise ValueError("FES computation requires at least one frame")
    w_host = None
    if weights is not None:
        w_host = np.asarray(weights, dtype=np.float64).reshape(-1)
    else:
        cand = split_data.get("weights") if isinstance(split_data, Mapping) else None
        if cand is None and isinstance(dataset.get("frame_weights"), Mapping):
            cand = dataset["frame_weights"].get(split_name)
        if cand is not None:
            w_host = np.asarray(cand, dtype=np.float64).reshape(-1)
    if w_host is not None and w_host.shape[0] != X.shape[0]:
        raise ValueError("Frame weights must match the number of frames in the split")
    method_norm = str(method or "kde").lower()
    if method_norm not in {"kde", "grid"}:
        raise ValueError("FES method must be either 'kde' or 'grid'")
    metadata: dict[str, Any] = {"temperature_K": float(temperature_K), "split": split_name,
                                "weighted": w_host is not None, "method": method_norm}

    eng = get_engine()
    n = X.shape[0]
    xd = eng.to_device(np.ascontiguousarray(X))
    _, std, _ = eng.column_moments(xd, ddof=0)
    if X.shape[1] < 2:
        raise ValueError(f"Coordinate array has {X.shape[1]} dimensions, need 2")
    sel = _pick_components(std.to_host() ** 2, 2)
    wd = eng.to_device(w_host) if w_host is not None else None
    total, ess, w_max = float(n), float(n), 1.0
    if wd is not None:  # validation and the normalisation constants from one device pass over w
        ws = eng.weighted_stats(wd, weights=wd)       # [sum w, sum w^2, sum w^2 / sum w, ., min, max]
        if not np.isfinite(ws[0]) or not np.isfinite(ws[5]) or ws[4] < 0.0:
            raise ValueError("Frame weights must be finite and non-negative")
        total = float(ws[0])
        if total <= 0.0:
            raise ValueError("Frame weights must sum to a positive value")
        sum_w2 = float(ws[2]) * total
        ess = total ** 2 / sum_w2 if sum_w2 > 0 else total
        w_max = float(ws[5])
    sx = eng.weighted_stats(xd, sel[0], wd)
    sy = eng.weighted_stats(xd, sel[1], wd)

    if method_norm == "kde":
        nx, ny = _bins_pair(bins)
        if nx < 2 or ny < 2:
            raise ValueError("KDE FES requires at least two bins per dimension")
        bw_x, bw_y = _bandwidth(float(sx[3]), ess, bandwidth), _bandwidth(float(sy[3]), ess, bandwidth)
        x_min, x_max = float(sx[4]) - 3.0 * bw_x, float(sx[5]) + 3.0 * bw_x
        y_min, y_max = float(sy[4]) - 3.0 * bw_y, float(sy[5]) + 3.0 * bw_y
        if not all(np.isfinite(v) for v in (x_min, x_max, y_min, y_max)) or x_min >= x_max or y_min >= y_max:
            raise ValueError("Coordinate range must be finite and strictly increasing")
        xedges = np.linspace(x_min, x_max, num=nx + 1, dtype=np.float64)
        yedges = np.linspace(y_min, y_max, num=ny + 1, dtype=np.float64)
        hist_d = eng.kde2d(xd, sel, 0.5 * (xedges[:-1] + xedges[1:]), 0.5 * (yedges[:-1] + yedges[1:]), bw_x, bw_y,
                           wd, 1.0 / total)
        metadata["bandwidth"] = {"selector": bandwidth, "x": bw_x, "y": bw_y, "effective_sample_size": ess,
                                 "total_weight": total}
        metadata["grid_shape"] = (nx, ny)
    else:
        nx, ny = _bins_pair(bins)
        xedges = _outer_edges(float(sx[4]), float(sx[5]), nx)
        yedges = _outer_edges(float(sy[4]), float(sy[5]), ny)
        hist_d = eng.hist2d(xd, sel, xedges, yedges, wd, w_max)
        smoothed = 0
        min_count = int(min_count_per_bin)
        if min_count > 0:
            raw_total = float(hist_d.to_host().sum())
            sm, smoothed = eng.smooth_sparse_bins(hist_d, min_count)
            if smoothed > 0:
                hist_d = sm
                if raw_total > 0:
                    eng.scale_to_total(hist_d, raw_total)
        metadata["min_count_per_bin"] = min_count
        metadata["smoothed_bins"] = smoothed
        if smoothed > 0:
            metadata["smoothing"] = "neighbor_average"
    metadata["selected_components"] = sel

    F_d, status = eng.fes_finalize(hist_d, BOLTZMANN_CONSTANT_KJ_PER_MOL * float(temperature_K))
    if status & 1:
        raise ValueError("Histogram must contain only finite values")
    if status & 2:
        raise ValueError("Histogram total must be positive and finite")
    if status & 4:
        raise ValueError("Histogram entries must be strictly positive for FES")
    F = F_d.to_host()
    if not np.all(np.isfinite(F)):
        raise FloatingPointError("Free energy computation produced non-finite values")
    return {"histogram": hist_d.to_host(), "xedges": xedges, "yedges": yedges, "free_energy": F, "metadata": metadata}
