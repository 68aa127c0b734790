"""Device-resident featurize -> TICA -> k-means -> counts -> T -> ITS pipeline.

This is the hot path itself (what bench.py times and what the operator shims in
``pmarlo_amd.features`` / ``.markov_state_model`` / ``.analysis`` call into): every
stage consumes and produces ``DeviceArray`` objects, nothing returns to the host
between stages, and the sequence can be captured into one hipGraph.

Stages and the reference operators they stand for:
  standardise  reduction._preprocess                      S/markov_state_model/reduction.py:13-40
  tica         deeptime TICA via tica_reduce              reduction.py:77-110, _features.py:181-231
  cluster      cluster_microstates / _KMeansDiscretizer   clustering.py:395-665, discretize.py:406-514
  count        _weighted_counts / deeptime sliding count  discretize.py:609-645, _estimation.py:116-156
  estimate     _normalise_counts / ML-MSM + pi            discretize.py:678-682, _estimation.py:158-188
  its          _summarize_its_stats on the ML matrix      _its.py:543-604
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Sequence

import numpy as np

from .device import DeviceArray, Engine, segments_to_bounds

__all__ = ["TicaModel", "MSMPipeline", "PipelineResult"]

_EPS_SCALE = 10 * np.finfo(float).eps  # sklearn _handle_zeros_in_scale


@dataclass
class TicaModel:
    """Fitted TICA on the device: z = (x - mu) / sigma; y = (z - mean) @ W[:, :dim]."""

    mu: DeviceArray
    sigma: DeviceArray          # divisor (1 where the column is constant or scale=False)
    inv_sigma: DeviceArray
    eigenvalues: DeviceArray    # [F]
    coefficients: DeviceArray   # [F, F]
    mean: DeviceArray           # symmetric mean in standardised coordinates
    rank: DeviceArray           # int32 [1]
    lag: int
    dim: int
    moments: DeviceArray | None = None


@dataclass
class PipelineResult:
    labels: DeviceArray
    centers: DeviceArray
    counts: DeviceArray
    transition_matrix: DeviceArray
    tica: TicaModel | None = None
    projected: DeviceArray | None = None
    extras: dict = field(default_factory=dict)


class MSMPipeline:
    """One shard of frames on one GPU.  Multi-GPU runs wrap the ``*_partial`` /
    ``*_finish`` pairs with an all-reduce of the small moment / count buffers
    (pmarlo_amd.dist)."""

    def __init__(self, engine: Engine):
        self.eng = engine

    # ---- standardisation moments --------------------------------------------
    def standardise_params(self, x: DeviceArray, scale: bool = True, sums: DeviceArray | None = None,
                           shift: DeviceArray | None = None, n_rows_total: int | None = None):
        """mean / divisor of reduction._preprocess: population std over ALL rows, where NaNs
        count as imputed means (they add 0 to the centred square sum but count in n)."""
        eng = self.eng
        n, F = x.shape
        if sums is None:
            sums, shift = eng.column_moments_partial(x)
        mean, std, cnt = eng.moments_finalize(sums, shift, F, ddof=0)
        # host touch of 3F doubles: the divisor needs the sklearn zero-scale rule
        std_h, cnt_h = std.to_host(), cnt.to_host()
        n_tot = float(n if n_rows_total is None else n_rows_total)
        std_imputed = std_h * np.sqrt(np.where(cnt_h > 0, cnt_h, 1.0) / n_tot)
        div = np.where(std_imputed < _EPS_SCALE, 1.0, std_imputed) if scale else np.ones(F)
        has_nan = bool(np.any(cnt_h < n_tot))
        return mean, eng.to_device(div), eng.to_device(1.0 / div), has_nan

    # ---- TICA ------------------------------------------------------------------
    def tica_moments(self, x: DeviceArray, lag: int, mu: DeviceArray, segments=None, assume_finite=False,
                     out: DeviceArray | None = None, symmetric: bool = False) -> DeviceArray:
        """Raw lagged moments over the segments; symmetric: M0t comes back as (M0t + M0t') / 2 from the cheaper
        symmetric accumulation (what a reversible TICA or a PCA needs; VAMP needs the plain block)."""
        starts, stops = segments_to_bounds(segments, x.shape[0])
        if len(starts) <= 16:
            return self.eng.lagged_moments(x, lag, mu, starts=starts, stops=stops, assume_finite=assume_finite, out=out,
                                           symmetric=symmetric)
        # more than 16 trajectories: accumulate the (additive) raw moments in chunks
        total = None
        for i in range(0, len(starts), 16):
            part = self.eng.lagged_moments(x, lag, mu, starts=starts[i:i + 16], stops=stops[i:i + 16],
                                           assume_finite=assume_finite, symmetric=symmetric).to_host()
            total = part if total is None else total + part
        return self.eng.to_device(total) if out is None else out.copy_from_host(total)

    def tica_fit(self, x: DeviceArray, lag: int, dim: int, *, scale: bool = True, segments=None,
                 epsilon: float = 1e-6, kinetic_map: bool = True) -> TicaModel:
        mu, sigma, inv_sigma, has_nan = self.standardise_params(x, scale=scale)
        mom = self.tica_moments(x, lag, mu, segments=segments, assume_finite=not has_nan, symmetric=True)
        return self.tica_solve(mom, mu, sigma, inv_sigma, lag, dim, epsilon=epsilon, kinetic_map=kinetic_map)

    def tica_solve(self, moments: DeviceArray, mu, sigma, inv_sigma, lag: int, dim: int, *, epsilon: float = 1e-6,
                   kinetic_map: bool = True) -> TicaModel:
        F = mu.shape[0]
        eig, W, mean, rank = self.eng.tica_solve(moments, F, scale=sigma, epsilon=epsilon, kinetic_map=kinetic_map)
        return TicaModel(mu, sigma, inv_sigma, eig, W, mean, rank, int(lag), int(dim), moments)

    def tica_transform(self, model: TicaModel, x: DeviceArray, out: DeviceArray | None = None,
                       assume_finite: bool = False) -> DeviceArray:
        return self.eng.project(x, model.mu, model.inv_sigma, model.coefficients, model.dim, mean2=model.mean, out=out,
                                assume_finite=assume_finite)

    # ---- clustering --------------------------------------------------------------
    def cluster(self, y: DeviceArray, k: int, *, seed: int = 0, max_iter: int = 50, tol: float = 1e-4,
                whiten: tuple[DeviceArray, DeviceArray] | None = None, centers: DeviceArray | None = None,
                labels: DeviceArray | None = None, mindist: DeviceArray | None = None, fit: bool = True):
        """Lloyd fit (unless centres are given with fit=False) + final assignment.
        tol follows sklearn: stop when the squared centre shift <= tol * mean column variance."""
        eng = self.eng
        mean, std = whiten if whiten is not None else (None, None)
        state = None
        if fit:
            n, d = y.shape
            if whiten is not None:
                tol2 = float(tol) * 1.0  # whitened columns have unit variance
            else:
                _, sd, _ = eng.column_moments(y, ddof=0)
                tol2 = float(tol) * float(np.mean(sd.to_host() ** 2))
            centers, state = eng.kmeans_fit(y, k, seed=seed, max_iter=max_iter, tol2=tol2, mean=mean, std=std,
                                            centers=centers)
        labels = eng.kmeans_assign(y, centers, mean=mean, std=std, labels=labels, mindist=mindist)
        return labels, centers, state

    # ---- counting + estimation ------------------------------------------------------
    def count(self, labels: DeviceArray, k: int, lag: int, segments=None, stride: int = 1, weights=None,
              out: DeviceArray | None = None, pairs: DeviceArray | None = None):
        starts, stops = segments_to_bounds(segments, labels.size)
        if weights is not None:
            return self.eng.count_transitions_weighted(labels, weights, k, lag, starts=starts, stops=stops,
                                                       stride=stride, out=out, pairs=pairs)
        return self.eng.count_transitions(labels, k, lag, starts=starts, stops=stops, stride=stride, out=out,
                                          pairs=pairs)

    def estimate(self, counts: DeviceArray, *, connected: bool = True, n_its: int = 0, lag: float = 1.0):
        """counts -> dict(T_active packed, active, n_active, pi, its...) on the device."""
        eng = self.eng
        out = eng.transition_matrix(counts, mode=1 if connected else 0)
        if connected:
            spec = eng.spectrum(out["T"], n=out["n_active"], n_its=n_its, lags=[lag])
            out.update(spectrum=spec, pi_active=spec["pi"])
        return out

    # ---- the whole path on one shard -----------------------------------------------
    def run(self, x: DeviceArray, *, lag: int, tica_dim: int, k: int, segments=None, seed: int = 0,
            kmeans_iter: int = 20, kmeans_tol: float = 1e-4, scale: bool = True) -> PipelineResult:
        model = self.tica_fit(x, lag, tica_dim, scale=scale, segments=segments)
        y = self.tica_transform(model, x)
        labels, centers, state = self.cluster(y, k, seed=seed, max_iter=kmeans_iter, tol=kmeans_tol)
        counts, pairs = self.count(labels, k, lag, segments=segments)
        tm = self.eng.transition_matrix(counts, mode=0)
        return PipelineResult(labels, centers, counts, tm["T"], model, y, {"pairs": pairs, "fit_state": state,
                                                                            "diag_mass": tm["diag_mass"]})
