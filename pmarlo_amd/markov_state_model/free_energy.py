"""Densities to free energies: the array-level half of pmarlo.markov_state_model.free_energy
(S/markov_state_model/free_energy.py:257-414): free_energy_from_density, periodic_kde_2d,
generate_1d_pmf.  The passes over the samples run on the device (wrapped-Gaussian KDE on the matrix
cores, histogram with np.histogram's edge rules); what is left on the host acts on grid-sized arrays.
generate_2d_fes (:417-868) is mirrored with its grid rules: the 1 % / 99 % crop (scipy mquantiles) and the
Freedman-Diaconis bin count (scipy iqr) take exact order statistics from the device (radix selection, no
sort), samples are clipped / wrapped and histogrammed on the device, the adaptive bin search re-histograms
there; masks, inpainting fill, the smoothing policies (fes_smoothing.py) and metadata are grid-sized host
logic.  Pinned by tests/golden/free_energy.npz and fes2d.npz, made by importing the reference module."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Mapping, Optional, Tuple

import numpy as np

from ..device import get_engine

__all__ = ["PMFResult", "FESResult", "FESCalculator", "kT_kJ_per_mol", "free_energy_from_density", "periodic_kde_2d", "generate_1d_pmf",
           "generate_2d_fes"]


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


class FESResult:
    """Free-energy surface with its edges and metadata (free_energy.py:43-160): `F` (alias `free_energy`),
    `xedges`, `yedges`, `levels_kJmol`, `metadata`; `counts`, `cv1_name`, `cv2_name`, `temperature` are mirrored
    between attributes and metadata."""

    version = "2.0"

    def __init__(self, F=None, *, free_energy=None, xedges, yedges, levels_kJmol=None, metadata=None, counts=None,
                 cv1_name=None, cv2_name=None, temperature=None) -> None:
        if F is None and free_energy is None:
            raise TypeError("FESResult requires either 'F' or 'free_energy' to be provided")
        if F is not None and free_energy is not None:
            import warnings

            warnings.warn("Both 'F' and 'free_energy' were provided; using 'F'", RuntimeWarning, stacklevel=2)
        self.F = np.asarray(F if F is not None else free_energy, dtype=np.float64)
        self.xedges = np.asarray(xedges, dtype=np.float64)
        self.yedges = np.asarray(yedges, dtype=np.float64)
        self.levels_kJmol = None if levels_kJmol is None else np.asarray(levels_kJmol, dtype=np.float64)
        meta = dict(metadata or {})
        c = counts if counts is not None else meta.get("counts")
        self.counts = None if c is None else np.asarray(c, dtype=np.float64)
        if self.counts is not None:
            meta["counts"] = self.counts
        self.cv1_name = cv1_name if cv1_name is not None else meta.get("cv1_name")
        self.cv2_name = cv2_name if cv2_name is not None else meta.get("cv2_name")
        for key, val in (("cv1_name", self.cv1_name), ("cv2_name", self.cv2_name)):
            if val is not None:
                meta.setdefault(key, val)
        t = temperature if temperature is not None else meta.get("temperature")
        self.temperature = None if t is None else float(t)
        if self.temperature is not None:
            meta["temperature"] = self.temperature
        self.metadata = meta

    @property
    def output_shape(self) -> tuple[int, int]:
        return int(self.F.shape[0]), int(self.F.shape[1])

    @property
    def free_energy(self) -> np.ndarray:
        return self.F

    def __getitem__(self, key: str):
        """Mapping-style access of older callers (``fes["F"]``): deprecated, as in the reference (free_energy.py:133-157)."""
        import warnings

        warnings.warn("Dictionary-style access to FESResult is deprecated; use attributes instead.", DeprecationWarning,
                      stacklevel=2)
        if key in ("F", "xedges", "yedges", "levels_kJmol"):
            return getattr(self, key)
        raise KeyError(key)

    # ---- the reference's wire format (free_energy.py:159-251): arrays as nested lists, or {"shape", "dtype"} stubs ----
    def to_dict(self, metadata_only: bool = False) -> dict:
        def plain(v):
            if isinstance(v, np.ndarray):
                return {"shape": list(v.shape), "dtype": str(v.dtype)} if metadata_only else v.tolist()
            return v

        out = {"version": self.version, "free_energy": plain(self.F), "xedges": plain(self.xedges), "yedges": plain(self.yedges)}
        for key in ("levels_kJmol", "counts"):
            if getattr(self, key) is not None:
                out[key] = plain(getattr(self, key))
        if self.temperature is not None:
            out["temperature"] = float(self.temperature)
        for key in ("cv1_name", "cv2_name"):
            if getattr(self, key) is not None:
                out[key] = getattr(self, key)
        # what lives in attributes is not repeated under "metadata"
        rest = {k: plain(v) for k, v in self.metadata.items() if k not in ("counts", "temperature", "cv1_name", "cv2_name")}
        if rest:
            out["metadata"] = rest
        return out

    @classmethod
    def from_dict(cls, data: dict) -> "FESResult":
        raw = dict(data)
        version = raw.pop("version", cls.version)
        if version not in ("1.0", "2.0"):
            raise ValueError(f"Version mismatch: {version} != {cls.version}")

        def back(v):
            if isinstance(v, dict) and "shape" in v and "dtype" in v:      # a metadata_only stub: zeros of that shape
                return np.zeros(tuple(int(x) for x in v["shape"]), dtype=np.dtype(v.get("dtype", "float64")))
            return np.asarray(v) if isinstance(v, list) else v

        extra = raw.pop("metadata", None) or {}
        temperature = raw.pop("temperature", None)
        opt = {key: raw.pop(key, None) for key in ("levels_kJmol", "counts")}
        return cls(F=back(raw.pop("free_energy")), xedges=back(raw.pop("xedges")), yedges=back(raw.pop("yedges")),
                   levels_kJmol=None if opt["levels_kJmol"] is None else back(opt["levels_kJmol"]),
                   counts=None if opt["counts"] is None else back(opt["counts"]),
                   metadata={k: back(v) for k, v in extra.items()},
                   cv1_name=raw.pop("cv1_name", None) or extra.get("cv1_name"),
                   cv2_name=raw.pop("cv2_name", None) or extra.get("cv2_name"),
                   temperature=temperature if temperature is not None else extra.get("temperature"))


def _reference_crop_is_live() -> bool:
    """The reference crops with ``mquantiles(x, prob=...).filled(np.nan)`` inside a try / except that falls back
    to the full data range (free_energy.py:497-547).  scipy's mquantiles returns a plain ndarray for plain
    input in current releases (1.15 here), so ``.filled`` raises and the crop silently never happens; with a
    scipy that hands back a masked array it does.  Results have to match the reference as it runs, so the same
    property of the installed scipy decides here."""
    from scipy.stats.mstats import mquantiles

    return hasattr(mquantiles(np.zeros(2), prob=[0.5]), "filled")


def _mquantile_pair(n: int, prob: float):
    """scipy.stats.mstats.mquantiles(alphap = betap = 0.4) position of one probability: (k - 1, k, gamma) with
    q = (1 - gamma) x_(k-1) + gamma x_(k), ranks 0-based."""
    aleph = n * prob + (0.4 + prob * 0.2)
    k = int(np.floor(min(max(aleph, 1), n - 1)))
    return k - 1, k, float(min(max(aleph - k, 0.0), 1.0))


def _percentile_pair(n: int, pct: float):
    """np.percentile (linear): position pct / 100 (n - 1) -> (lo, hi, tag); the interpolation weight t is passed as
    tag = -(t + 1) so that order_stats applies numpy's lerp rule to it rather than the mquantiles blend."""
    pos = pct / 100.0 * (n - 1)
    lo = int(np.floor(pos))
    return lo, min(lo + 1, n - 1), -(float(pos - lo) + 1.0)


def _fd_bin_count(q25: float, q75: float, n: int, span: float, eps: float) -> int:
    """Freedman-Diaconis bins over `span` (8 .. 512), 0 when the rule does not apply (:575-590)."""
    if n <= 1 or not np.isfinite(span) or span <= 0:
        return 0
    width = 2.0 * (q75 - q25) / np.cbrt(max(1, n))
    if not np.isfinite(width) or width <= eps:
        return 0
    nb = int(np.ceil(span / width))
    return 0 if nb <= 0 else int(np.clip(nb, 8, 512))


def generate_2d_fes(cv1, cv2, bins: Tuple[int, int] = (100, 100), temperature: float = 300.0,  # noqa: C901
                    periodic: Tuple[bool, bool] = (False, False), ranges=None, min_count: int = 1,
                    kde_bw_deg: Tuple[float, float] = (20.0, 20.0), epsilon: float = 1e-6, config=None,
                    grid_strategy: str = "adaptive", fes_smoothing_mode: str | None = None) -> FESResult:
    """Two-dimensional free-energy surface from samples (free_energy.py:417-868): 1 % / 99 % crop of
    non-periodic data, Freedman-Diaconis / sqrt(N) / requested bin counts, for the adaptive strategy a search
    for a grid with >= 60 % populated bins, periodic axes wrapped into their range, bins below min_count masked,
    optional uncertainty-driven smoothing (mode "never" | "auto" | "always", fes_* entries of `config`)."""
    from .fes_smoothing import adaptive_bandwidth, beta_to_kT, mark_bins_for_smoothing, smooth_F_with_adaptive_gaussian

    x = np.asarray(cv1, dtype=np.float64).reshape(-1)
    y = np.asarray(cv2, dtype=np.float64).reshape(-1)
    if x.size == 0 or y.size == 0:
        raise ValueError("cv1 and cv2 must not be empty")
    if x.shape != y.shape:
        raise ValueError("cv1 and cv2 must have the same shape")
    if len(bins) != 2 or any(b <= 0 for b in bins):
        raise ValueError("bins must be a tuple of two positive integers")
    if temperature <= 0:
        raise ValueError("temperature must be positive")
    if len(periodic) != 2:
        raise ValueError("periodic must be a tuple of two booleans")
    if min_count < 0:
        raise ValueError("min_count must be non-negative")
    grid_strategy = str(grid_strategy).lower()
    if grid_strategy not in {"fixed", "adaptive"}:
        raise ValueError("grid_strategy must be 'fixed' or 'adaptive'")
    eng = get_engine()
    n = int(x.size)
    xy = eng.to_device(np.ascontiguousarray(np.stack([x, y], axis=1)))
    crop = ranges is None and not any(periodic) and _reference_crop_is_live()
    if ranges is not None and (len(ranges) != 2 or any(len(r) != 2 for r in ranges)):
        raise ValueError("ranges must be ((xmin, xmax), (ymin, ymax))")

    def order_stats(source, col, pairs, clip=None):
        """Interpolated order statistics (1 - g) x_(lo) + g x_(hi); `clip` is applied to the order statistics
        first (the statistics of clipped samples are the clipped statistics: clipping is monotone)."""
        ranks = sorted({int(r) for lo, hi, _ in pairs for r in (lo, hi)})
        vals = eng.order_statistics(source, ranks, col=col)
        if clip is not None:
            vals = np.clip(vals, clip[0], clip[1])
        table = dict(zip(ranks, vals))
        out = []
        for lo, hi, g in pairs:
            a, b = table[lo], table[hi]
            if g < 0:                                       # np.percentile's lerp (t = -g - 1 was tagged by the caller)
                t = -g - 1.0
                out.append(a + (b - a) * t if t < 0.5 else b - (b - a) * (1.0 - t))
            else:                                           # mquantiles: (1 - gamma) x_(k-1) + gamma x_(k)
                out.append((1.0 - g) * a + g * b)
        return out

    limits, quart, axes = [], [], []
    for axis in range(2):
        st = eng.weighted_stats(xy, axis)
        lo, hi = float(st[4]), float(st[5])
        if ranges is not None:
            lo, hi = float(ranges[axis][0]), float(ranges[axis][1])
        elif crop:
            q01, q99 = order_stats(xy, axis, [_mquantile_pair(n, 0.01), _mquantile_pair(n, 0.99)])
            if np.isfinite(q01) and np.isfinite(q99) and q99 > q01:
                lo, hi = float(q01), float(q99)
        limits.append((lo, hi))
    (xlo, xhi), (ylo, yhi) = limits
    if not np.isfinite([xlo, xhi, ylo, yhi]).all() or xlo >= xhi or ylo >= yhi:
        raise ValueError("ranges must be finite with min < max for both axes")
    quartile_pairs = [_percentile_pair(n, 25.0), _percentile_pair(n, 75.0)]
    for axis, (lo, hi) in enumerate(limits):
        if periodic[axis]:                                  # wrapped into [lo, hi): not monotone, select on the copy
            wrapped = eng.clip_or_wrap(xy, lo, hi, wrap=True, col=axis)
            axes.append(wrapped)
            quart.append(order_stats(wrapped, 0, quartile_pairs))
        elif crop:                                          # clipped into [lo, hi] to keep the edge bins populated
            axes.append(eng.clip_or_wrap(xy, lo, hi, wrap=False, col=axis))
            quart.append(order_stats(xy, axis, quartile_pairs, clip=(lo, hi)))
        else:                                               # given range: samples outside simply fall off the grid
            axes.append((xy, axis))
            quart.append(order_stats(xy, axis, quartile_pairs))
    dev = axes
    sqrt_bins = max(8, int(np.sqrt(max(1, n))))
    fd = [_fd_bin_count(quart[a][0], quart[a][1], n, limits[a][1] - limits[a][0], float(epsilon)) for a in range(2)]
    bx = max(int(bins[0]), fd[0], sqrt_bins)
    by = max(int(bins[1]), fd[1], sqrt_bins)
    if grid_strategy == "adaptive":
        for _ in range(10):
            H = eng.hist2d_xy(dev[0], dev[1], np.linspace(xlo, xhi, bx + 1), np.linspace(ylo, yhi, by + 1)).to_host()
            if 1.0 - float(np.sum(H < min_count)) / H.size >= 0.6:
                break
            bx, by = max(8, int(bx * 0.75)), max(8, int(by * 0.75))
    xedges = np.linspace(xlo, xhi, bx + 1)
    yedges = np.linspace(ylo, yhi, by + 1)
    xh = np.concatenate([xedges, [xedges[-1] + (xedges[1] - xedges[0])]]) if periodic[0] else xedges
    yh = np.concatenate([yedges, [yedges[-1] + (yedges[1] - yedges[0])]]) if periodic[1] else yedges
    H = eng.hist2d_xy(dev[0], dev[1], xh, yh).to_host()
    if periodic[0]:
        H[0, :] += H[-1, :]
        H = H[:-1, :]
    if periodic[1]:
        H[:, 0] += H[:, -1]
        H = H[:, :-1]
    area = np.diff(xedges)[0] * np.diff(yedges)[0]
    empty = H < min_count
    total = float(H.sum())
    if total <= 0.0:
        raise ValueError("Histogram counts sum to zero; cannot compute FES")
    density = H / (total * area)
    F_masked = free_energy_from_density(density, temperature, mask=empty, inpaint=False)
    finite = np.isfinite(F_masked)
    if finite.any():
        F_numeric = np.where(finite, F_masked, float(np.nanmax(F_masked[finite])))
    else:
        F_numeric = free_energy_from_density(density, temperature, mask=None, inpaint=True)
        if not np.isfinite(F_numeric).any():
            raise ValueError("No finite free-energy values available for smoothing")

    def cfg(name, default):
        if config is None:
            return default
        if isinstance(config, Mapping):
            return config.get(name, default)
        return getattr(config, name, default)

    mode_cfg = fes_smoothing_mode if fes_smoothing_mode is not None else cfg("fes_smoothing_mode", None)
    mode = str(mode_cfg).lower() if mode_cfg is not None else "never"
    if mode not in {"never", "auto", "always"}:
        raise ValueError(f"Unknown fes_smoothing_mode={mode!r}")
    target = cfg("fes_target_sd_kT", None)
    target = 0.5 if target is None else float(target)
    alpha, h0, ess_ref = float(cfg("fes_alpha", 1e-6)), float(cfg("fes_h0", 1.2)), float(cfg("fes_ess_ref", 50.0))
    h_min, h_max = float(cfg("fes_h_min", 0.4)), float(cfg("fes_h_max", 3.0))
    if alpha <= 0:
        raise ValueError("fes_alpha must be positive")
    if h0 <= 0:
        raise ValueError("fes_h0 must be positive")
    if ess_ref <= 0:
        raise ValueError("fes_ess_ref must be positive")
    if h_min <= 0 or h_max <= 0 or h_min > h_max:
        raise ValueError("fes_h_min and fes_h_max must be positive with h_min <= h_max")
    sd_map = bw_map = None
    smooth_mask = np.zeros_like(empty, dtype=bool)
    if mode == "never":
        F_smooth, applied = F_numeric, np.zeros_like(empty, dtype=bool)
    else:
        smooth_mask, sd_map = mark_bins_for_smoothing(H, target_sd_kT=target, alpha=alpha,
                                                      kT=beta_to_kT(1.0 / kT_kJ_per_mol(float(temperature))))
        smooth_mask = np.asarray(smooth_mask, dtype=bool)
        bw_map = adaptive_bandwidth(H.astype(float), h0=h0, ess_ref=ess_ref, h_min=h_min, h_max=h_max)
        F_smooth = smooth_F_with_adaptive_gaussian(F_numeric, h_map=bw_map, apply_mask=smooth_mask if mode == "auto" else None)
        applied = np.ones_like(empty, dtype=bool) if mode == "always" else smooth_mask
    final_mask = np.asarray(empty & ~applied, dtype=bool)
    F = np.where(final_mask, np.nan, F_smooth)
    if np.isfinite(F).any():
        F = F - float(np.nanmin(F[np.isfinite(F)]))
    empty_fraction = float(np.count_nonzero(empty)) / float(H.size)
    smoothing = {"mode": mode, "target_sd_kT": target, "alpha": alpha, "h0": h0, "ess_ref": ess_ref, "h_min": h_min,
                 "h_max": h_max, "applied_fraction": float(np.mean(applied.astype(float)))}
    if sd_map is not None:
        smoothing["sd_map_kT"] = sd_map
    if bw_map is not None:
        smoothing["bandwidth_map"] = bw_map
    if smooth_mask.any():
        smoothing["mask"] = smooth_mask
    metadata = {"counts": density, "periodic": periodic, "temperature": temperature, "mask": final_mask,
                "empty_bins_fraction": empty_fraction, "smoothing": smoothing, "grid_strategy": grid_strategy,
                "grid_shape": (bx, by), "grid_ranges": {"x": (xlo, xhi), "y": (ylo, yhi)}}
    if empty_fraction > 0.50:
        metadata["sparse_warning"] = (f"Sparse FES: {empty_fraction * 100.0:.1f}% empty bins detected. "
                                      f"Grid: {bx}\u00d7{by}. Consider using grid_strategy='adaptive' to reduce waste.")
    return FESResult(F=F, xedges=xedges, yedges=yedges, metadata=metadata)


class FESCalculator:
    """MSM-reweighted free-energy surface in kT units (free_energy.py:867-1060): every frame of the projected
    trajectories is weighted by the stationary probability of its microstate, the weighted density histogram
    (np.histogram2d(..., weights, density=True) over the data range) becomes -ln(density), shifted to zero and
    optionally capped.  Weights gather, range and histogram run on the device."""

    def __init__(self, config: dict):
        self.config = config
        self.temperature = config.get("temperature", 300.0)
        self.kbt = 0.00831446261815324 * self.temperature

    def calculate_fes(self, projection, msm, dtrajs=None, bins: int = 150, max_energy_cap_kt: Optional[float] = 10.0,
                      dim_x: int = 0, dim_y: int = 1):
        """-> ([xx, yy] bin-centre meshgrids, F in kT), or (None, None) when the inputs do not fit together."""
        if not projection or msm is None or not hasattr(msm, "stationary_distribution"):
            return None, None
        if dtrajs is None:
            dtrajs = getattr(msm, "discrete_trajectories", None)
            if dtrajs is None:
                dtrajs = getattr(msm, "_dtrajs", None)
            if dtrajs is None:
                return None, None
        if isinstance(dtrajs, np.ndarray):
            dtrajs = [dtrajs]
        try:
            labels = np.concatenate([np.asarray(d, dtype=int).reshape(-1) for d in dtrajs])
            if projection[0].shape[1] <= max(dim_x, dim_y):
                return None, None
            xy = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64)[:, [dim_x, dim_y]] for p in projection]))
            pi_raw = getattr(msm, "stationary_distribution", None)
            if pi_raw is None:
                return None, None
            pi = np.asarray(pi_raw, dtype=float)
            if pi.size == 0 or xy.shape[0] != labels.size:
                return None, None
            if labels.max() >= pi.size:                       # frames of states the model does not know are dropped
                keep = labels < pi.size
                xy, labels = np.ascontiguousarray(xy[keep]), labels[keep]
                if labels.size == 0:
                    return None, None
            eng = get_engine()
            xd = eng.to_device(xy)
            w = eng.gather(eng.to_device(pi), eng.to_device(labels.astype(np.int32)))
            sx, sy = eng.weighted_stats(xd, 0), eng.weighted_stats(xd, 1)
            xe = np.linspace(float(sx[4]), float(sx[5]), int(bins) + 1)
            ye = np.linspace(float(sy[4]), float(sy[5]), int(bins) + 1)
            hist = eng.hist2d(xd, (0, 1), xe, ye, weights=w, w_absmax=float(pi.max())).to_host()
            hist = hist / hist.sum() / np.outer(np.diff(xe), np.diff(ye))          # density=True
            xx, yy = np.meshgrid(0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:]))
            F = -self.kbt * np.log(np.maximum(hist, np.finfo(hist.dtype).tiny))
            F = (F - F.min()) / self.kbt
            if max_energy_cap_kt is not None:
                F = np.clip(F, 0, max_energy_cap_kt)
            return [xx, yy], F
        except Exception:
            return None, None
