"""TICA reduction: mirror of pmarlo.markov_state_model.reduction
(S/markov_state_model/reduction.py:13-40 _preprocess, :77-110 tica_reduce, :152 reduce_features)
and of FeaturesMixin._maybe_apply_tica (S/markov_state_model/_features.py:181-231)."""

from __future__ import annotations

from typing import List, Sequence

import numpy as np

from ..device import get_engine
from ..pipeline import MSMPipeline, TicaModel

__all__ = ["tica_reduce", "reduce_features", "tica_fit_transform_trajectories", "preprocess_params"]


def _as_matrix(X) -> np.ndarray:
    X = np.asarray(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    if X.ndim == 1:
        X = X.reshape(-1, 1)
    if X.ndim != 2:
        raise ValueError(f"Expected 2D array, got shape {X.shape}")
    return np.ascontiguousarray(X)


def preprocess_params(X, scale: bool = True):
    """(mean, divisor) of _preprocess (imputed-mean NaNs, population std, zero std -> 1)."""
    eng = get_engine()
    pipe = MSMPipeline(eng)
    mu, sigma, _, _ = pipe.standardise_params(eng.to_device(_as_matrix(X)), scale=scale)
    return mu.to_host(), sigma.to_host()


def tica_reduce(X: np.ndarray, lag: int = 1, n_components: int = 2, scale: bool = True) -> np.ndarray:
    """TICA(lagtime=lag, dim=n_components).fit([X_prep]).transform(X_prep) -> (N, d) float64."""
    Xm = _as_matrix(X)
    if Xm.size == 0:
        return np.zeros((Xm.shape[0], 0))
    eng = get_engine()
    pipe = MSMPipeline(eng)
    xd = eng.to_device(Xm)
    model = pipe.tica_fit(xd, int(lag), int(n_components), scale=scale)
    rank = int(model.rank.to_host()[0])
    if rank == 0:
        raise ValueError("TICA: covariance matrix has zero rank")
    model.dim = min(int(n_components), rank)
    return np.ascontiguousarray(pipe.tica_transform(model, xd).to_host(), dtype=float)


def reduce_features(X: np.ndarray, method: str = "tica", n_components: int = 2, lag: int = 1, scale: bool = True,
                    **kwargs) -> np.ndarray:
    method = method.lower()
    if method == "tica":
        return tica_reduce(X, lag=lag, n_components=n_components, scale=scale, **kwargs)
    if method in ("pca", "vamp"):
        raise NotImplementedError(f"{method!r} is outside the accelerated path (TICA only)")
    raise ValueError(f"Unknown reduction method: {method}")


def tica_fit_transform_trajectories(features: np.ndarray, traj_lengths: Sequence[int], n_components_hint: int,
                                    lag: int) -> tuple[np.ndarray, TicaModel]:
    """_maybe_apply_tica: dims clamped to [2, 5], fit on the list of trajectories (no
    standardisation there), transform each, drop the last ``lag`` frames of each, vstack."""
    n_components = int(max(2, min(5, n_components_hint)))
    drop = int(max(0, lag or 0))          # lag 0: fitted at lag 1, nothing dropped (_features.py:200, 216-231)
    lag = int(max(1, lag or 1))
    Xm = _as_matrix(features)
    edges = np.concatenate([[0], np.cumsum([int(v) for v in traj_lengths])])
    if edges[-1] != Xm.shape[0]:
        raise ValueError("trajectory lengths do not add up to the number of feature rows")
    segs = [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:])]
    eng = get_engine()
    pipe = MSMPipeline(eng)
    xd = eng.to_device(Xm)
    # deeptime removes the data mean itself; no scaling: sigma = 1
    mu, _, _, has_nan = pipe.standardise_params(xd, scale=False)
    one = eng.to_device(np.ones(Xm.shape[1]))
    mom = pipe.tica_moments(xd, lag, mu, segments=segs, assume_finite=not has_nan)
    model = pipe.tica_solve(mom, mu, one, one, lag, n_components)
    model.dim = min(n_components, int(model.rank.to_host()[0]))
    Y = pipe.tica_transform(model, xd).to_host()
    keep: List[np.ndarray] = []
    for a, b in segs:
        keep.append(Y[a:b - drop] if b - a > drop else np.empty((0, Y.shape[1])))
    return (np.vstack(keep) if keep else Y), model
