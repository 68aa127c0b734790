"""Densities to free energies: the array-level half of pmarlo.markov_state_model.free_energy
(S/markov_state_model/free_energy.py:257-414): free_energy_from_density, periodic_kde_2d,
generate_1d_pmf.  The passes over the samples run on the device (wrapped-Gaussian KDE on the matrix
cores, histogram with np.histogram's edge rules); what is left on the host acts on grid-sized arrays.
Pinned by tests/golden/free_energy.npz, made by importing the reference module.  The large
generate_2d_fes driver (adaptive grids, inpainting, smoothing policies) is not mirrored: its
building blocks are compute_weighted_fes (analysis/fes.py) and the functions here."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from ..device import get_engine

__all__ = ["PMFResult", "kT_kJ_per_mol", "free_energy_from_density", "periodic_kde_2d", "generate_1d_pmf"]


@dataclass
class PMFResult:
    F: np.ndarray
    edges: np.ndarray
    counts: np.ndarray
    periodic: bool
    temperature: float

    @property
    def output_shape(self) -> tuple[int, ...]:
        return tuple(int(v) for v in self.F.shape)


def kT_kJ_per_mol(temperature_kelvin: float) -> float:
    """k_B T N_A / 1000 with the CODATA 2018 exact constants (S/utils/thermodynamics.py:8-24)."""
    return 1.380649e-23 * float(temperature_kelvin) * 6.02214076e23 / 1000.0


def free_energy_from_density(density, temperature: float, *, mask=None, inpaint: bool = False,
                             tiny: float | None = None) -> np.ndarray:
    """F = -kT ln(density), +inf where the density does not exceed `tiny`, NaN where `mask` is set (unless the
    caller inpainted those bins), shifted so the lowest finite value is 0 (free_energy.py:257-310)."""
    if temperature <= 0:
        raise ValueError("temperature must be positive when computing free energy")
    rho = np.asarray(density, dtype=np.float64)
    floor = float(np.finfo(np.float64).tiny if tiny is None else tiny)
    with np.errstate(divide="ignore", invalid="ignore"):
        F = np.where(rho > floor, -kT_kJ_per_mol(float(temperature)) * np.log(np.clip(rho, floor, None)), np.inf)
    if mask is not None and not inpaint:
        F = np.where(np.asarray(mask, dtype=bool), np.nan, F)
    if np.any(np.isfinite(F)):
        F = F - np.nanmin(F)
    return F


def periodic_kde_2d(theta_x, theta_y, bw: Tuple[float, float] = (0.35, 0.35),
                    gridsize: Tuple[int, int] = (42, 42)) -> np.ndarray:
    """Wrapped-Gaussian kernel density on the torus [-pi, pi)^2, evaluated on linspace(-pi, pi, g, endpoint=False)
    per axis: sum_n exp(-(wrap(x_i - x_n) / s_x)^2 / 2) exp(-(wrap(y_j - y_n) / s_y)^2 / 2) / (N 2 pi s_x s_y)
    (free_energy.py:321-360).  One fp64 matrix-core pass over the samples."""
    x = np.asarray(theta_x, dtype=np.float64).reshape(-1)
    y = np.asarray(theta_y, dtype=np.float64).reshape(-1)
    if x.size == 0 or y.size == 0:
        raise ValueError("theta_x and theta_y must not be empty")
    if x.shape != y.shape:
        raise ValueError("theta_x and theta_y must have the same shape")
    sx, sy = float(bw[0]), float(bw[1])
    if sx <= 0 or sy <= 0:
        raise ValueError("bandwidth components must be positive")
    gx, gy = int(gridsize[0]), int(gridsize[1])
    if gx <= 0 or gy <= 0:
        raise ValueError("gridsize must be positive")
    eng = get_engine()
    xy = eng.to_device(np.ascontiguousarray(np.stack([x, y], axis=1)))
    dens = eng.kde2d(xy, (0, 1), np.linspace(-np.pi, np.pi, gx, endpoint=False),
                     np.linspace(-np.pi, np.pi, gy, endpoint=False), sx, sy, None, 1.0 / x.size, periodic=3)
    return dens.to_host()


def generate_1d_pmf(cv, bins: int = 100, temperature: float = 300.0, periodic: bool = False,
                    range_: Optional[Tuple[float, float]] = None, smoothing_sigma: Optional[float] = None) -> PMFResult:
    """np.histogram(cv, bins, range, density=True) on the device, optional Gaussian smoothing of the bins
    (scipy, wrap / reflect), free energy of the density (free_energy.py:363-414)."""
    from scipy.ndimage import gaussian_filter

    cv = np.asarray(cv, dtype=float).reshape(-1)
    if cv.size == 0:
        raise ValueError("cv array must not be empty")
    if bins <= 0:
        raise ValueError("bins must be positive")
    if temperature <= 0:
        raise ValueError("temperature must be positive")
    if smoothing_sigma is not None and smoothing_sigma < 0:
        raise ValueError("smoothing_sigma must be non-negative")
    eng = get_engine()
    xd = eng.to_device(np.ascontiguousarray(cv.reshape(-1, 1)))
    if range_ is None:
        st = eng.weighted_stats(xd, 0)
        lo, hi = float(st[4]), float(st[5])
        if not np.isfinite(st[2]):
            lo = hi = float("nan")
    else:
        lo, hi = float(range_[0]), float(range_[1])
    if not (np.isfinite(lo) and np.isfinite(hi)) or lo >= hi:
        raise ValueError("range_ must be finite with min < max")
    edges = np.linspace(lo, hi, int(bins) + 1)
    counts = eng.hist2d(xd, (0, 0), edges, np.asarray([lo, hi])).to_host().reshape(-1)
    H = counts / counts.sum() / np.diff(edges)
    if smoothing_sigma and smoothing_sigma > 0:
        H = gaussian_filter(H, sigma=float(smoothing_sigma), mode="wrap" if periodic else "reflect")
    return PMFResult(F=free_energy_from_density(H, temperature), edges=edges, counts=H, periodic=periodic,
                     temperature=temperature)
