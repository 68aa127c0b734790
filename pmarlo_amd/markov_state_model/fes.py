"""MSM-reweighted free-energy surface of two collective variables: the numerics of FESMixin
(S/markov_state_model/_fes.py:67-115 generate_free_energy_surface, :132-153 frame weights, :215-262 weighted
histogram and free energy, :283-291 bin choice) as functions on arrays.

Frame weights are pi[state] (the reference asks deeptime for trajectory weights first and falls back to this
when that fails; deeptime is absent here, so the fall-back is the behaviour that could be pinned).  Weight
gather, the weighted histogram over the frames and its edge rules run on the device; the Gaussian smoothing of
the bins (scipy, wrap / reflect) and the free energy act on the grid.  Pinned by tests/golden/msm_fes.npz,
made by calling the reference's mixin methods."""

from __future__ import annotations

from typing import Any, Dict, Optional, Sequence

import numpy as np

from ..device import get_engine

__all__ = ["choose_bins", "stationary_frame_weights", "weighted_density_histogram", "histogram_to_free_energy",
           "generate_free_energy_surface"]

_MIN_POSITIVE = 1e-12      # constants.NUMERIC_MIN_POSITIVE


def choose_bins(total_frames: int, user_bins: int) -> int:
    """40 .. 60 bins: sqrt(frames) // 6 unless the request (clamped to the same range) is within 5 of it."""
    if total_frames <= 0:
        return max(40, min(60, user_bins))
    reco = int(max(40, min(60, np.sqrt(total_frames) // 6)))
    cand = max(40, min(60, int(user_bins)))
    return cand if abs(cand - reco) <= 5 else reco


def stationary_frame_weights(dtrajs: Sequence[np.ndarray], stationary_distribution) -> np.ndarray:
    """pi[state] for every frame, trajectories concatenated (device gather)."""
    if stationary_distribution is None:
        raise ValueError("Stationary distribution not available")
    labels = np.concatenate([np.asarray(d, dtype=np.int64).reshape(-1) for d in dtrajs]) if len(dtrajs) else np.zeros(0, int)
    pi = np.ascontiguousarray(stationary_distribution, dtype=np.float64)
    if labels.size == 0:
        return np.zeros(0)
    if labels.min() < -pi.size or labels.max() >= pi.size:
        raise IndexError("state index out of range of the stationary distribution")
    eng = get_engine()
    return eng.gather(eng.to_device(pi), eng.to_device(np.mod(labels, pi.size).astype(np.int32))).to_host()


def weighted_density_histogram(cv1, cv2, weights, bins: int, ranges=None, smooth_sigma: Optional[float] = None,
                               periodic: bool = False):
    """np.histogram2d(cv1, cv2, bins, range, weights, density=True) on the device, then an optional Gaussian
    filter over the bins (mode "wrap" for periodic variables, else "reflect").  -> (H, xedges, yedges)."""
    from scipy.ndimage import gaussian_filter

    try:
        x = np.asarray(cv1, dtype=np.float64).reshape(-1)
        y = np.asarray(cv2, dtype=np.float64).reshape(-1)
        w = np.asarray(weights, dtype=np.float64).reshape(-1)
        if not (x.size == y.size == w.size) or x.size == 0:
            raise ValueError("cv1, cv2 and the weights must be non-empty and of one length")
        eng = get_engine()
        xy = eng.to_device(np.ascontiguousarray(np.stack([x, y], axis=1)))
        if ranges is None:
            sx, sy = eng.weighted_stats(xy, 0), eng.weighted_stats(xy, 1)
            ranges = [(float(sx[4]), float(sx[5])), (float(sy[4]), float(sy[5]))]
        xe = np.linspace(float(ranges[0][0]), float(ranges[0][1]), int(bins) + 1)
        ye = np.linspace(float(ranges[1][0]), float(ranges[1][1]), int(bins) + 1)
        H = eng.hist2d(xy, (0, 1), xe, ye, weights=eng.to_device(w), w_absmax=float(np.abs(w).max())).to_host()
        H = H / H.sum() / np.outer(np.diff(xe), np.diff(ye))
        if smooth_sigma and smooth_sigma > 0:
            H = gaussian_filter(H, sigma=float(smooth_sigma), mode="wrap" if periodic else "reflect")
        return H, xe, ye
    except Exception as exc:
        raise ValueError(f"Could not generate histogram for FES: {exc}")


def histogram_to_free_energy(H: np.ndarray, temperature: float) -> np.ndarray:
    """F = -kT ln H where H > 1e-12 (inf elsewhere), lowest value shifted to 0."""
    kT = 1.380649e-23 * temperature * 6.02214076e23 / 1000.0
    H = np.asarray(H, dtype=float)
    F = np.full_like(H, np.inf)
    ok = H > _MIN_POSITIVE
    if not ok.any():
        raise ValueError("Histogram too sparse for free energy calculation. Try fewer bins or more data")
    F[ok] = -kT * np.log(H[ok])
    F[np.isfinite(F)] -= float(np.min(F[np.isfinite(F)]))
    return F


def generate_free_energy_surface(cv1, cv2, dtrajs: Sequence[np.ndarray], stationary_distribution, *,
                                 cv1_name: str = "phi", cv2_name: str = "psi", bins: int = 50,
                                 temperature: float = 300.0) -> Dict[str, Any]:
    """The FES dictionary FESMixin stores: frames weighted by pi[state], 40 .. 60 bins, the (-180, 180) square
    with wrap-around smoothing for phi / psi (degrees), sigma = 0.6 bins."""
    w = stationary_frame_weights(dtrajs, stationary_distribution)
    x = np.asarray(cv1, dtype=float).reshape(-1)
    y = np.asarray(cv2, dtype=float).reshape(-1)
    nb = choose_bins(int(w.size), bins)
    m = min(x.size, y.size, w.size)
    if x.size != w.size:                       # _align_data_lengths: truncate to the common length
        x, y, w = x[:m], y[:m], w[:m]
    torsions = cv1_name == "phi" and cv2_name == "psi"
    H, xe, ye = weighted_density_histogram(x, y, w, nb, [(-180.0, 180.0), (-180.0, 180.0)] if torsions else None,
                                           smooth_sigma=0.6, periodic=torsions)
    return {"free_energy": histogram_to_free_energy(H, temperature), "xedges": xe, "yedges": ye, "cv1_name": cv1_name,
            "cv2_name": cv2_name, "temperature": temperature}
