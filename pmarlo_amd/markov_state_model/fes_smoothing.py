"""Uncertainty-driven smoothing of a free-energy grid: mirror of pmarlo.markov_state_model.fes_smoothing
(S/markov_state_model/fes_smoothing.py:6-110).  Everything here acts on the bins of an already-built grid
(host arithmetic, scipy filters as in the reference); the passes over the samples are in free_energy.py."""

from __future__ import annotations

import numpy as np

__all__ = ["beta_to_kT", "fes_uncertainty_sd_kT", "mark_bins_for_smoothing", "adaptive_bandwidth",
           "smooth_F_with_adaptive_gaussian"]


def beta_to_kT(beta: float) -> float:
    if beta <= 0:
        raise ValueError("beta must be > 0")
    return 1.0 / beta


def fes_uncertainty_sd_kT(bin_counts, alpha: float = 1e-6, kT: float = 1.0) -> np.ndarray:
    """SD of F_i = -kT ln p_i under a Dirichlet(n + alpha) posterior: kT sqrt(psi1(n_i + alpha) + psi1(N + K alpha))."""
    from scipy.special import polygamma

    n = np.asarray(bin_counts, dtype=float)
    if np.any(n < 0):
        raise ValueError("bin_counts must be non-negative")
    return kT * np.sqrt(polygamma(1, n + alpha) + polygamma(1, float(n.sum()) + alpha * n.size))


def mark_bins_for_smoothing(bin_counts, target_sd_kT: float = 0.5, alpha: float = 1e-6, kT: float = 1.0):
    sd = fes_uncertainty_sd_kT(bin_counts, alpha=alpha, kT=kT)
    return sd > float(target_sd_kT), sd


def adaptive_bandwidth(ess_map, h0: float = 1.2, ess_ref: float = 50.0, h_min: float = 0.4, h_max: float = 3.0,
                       eps: float = 1e-12) -> np.ndarray:
    """h = h0 sqrt(ess_ref / max(ESS, eps)) clipped to [h_min, h_max]."""
    ess = np.maximum(np.asarray(ess_map, dtype=float), eps)
    return np.clip(h0 * np.sqrt(ess_ref / ess), h_min, h_max)


def smooth_F_with_adaptive_gaussian(F, h_map, apply_mask=None, sigma_grid=(0.5, 1.0, 2.0, 3.0)) -> np.ndarray:
    """Per-bin Gaussian width by linear interpolation between a few global blurs (scipy gaussian_filter,
    mode "nearest"): a bin with width h takes (1 - w) blur[s_lo] + w blur[s_hi] for the grid widths around h."""
    from scipy.ndimage import gaussian_filter

    F = np.asarray(F, dtype=float)
    h_map = np.asarray(h_map, dtype=float)
    if F.shape != h_map.shape:
        raise ValueError("F and h_map must have the same shape")
    sig = np.asarray([float(v) for v in sigma_grid])
    blurred = np.stack([gaussian_filter(F, sigma=v, mode="nearest") for v in sig])
    h = np.clip(h_map, sig[0], sig[-1])
    hi = np.clip(np.searchsorted(sig, h, side="right"), 1, sig.size - 1)
    lo = hi - 1
    w = (h - sig[lo]) / np.maximum(sig[hi] - sig[lo], 1e-12)
    pick = lambda idx: np.take_along_axis(blurred, idx[np.newaxis, ...], axis=0)[0]  # noqa: E731
    out = (1.0 - w) * pick(lo) + w * pick(hi)
    return out if apply_mask is None else np.where(apply_mask, out, F)
