"""Seeded synthetic inputs shared by tests and bench.py (recipes follow the
reference's perf tests)."""

from __future__ import annotations

import numpy as np


def correlated_series(n_frames: int, n_features: int, seed: int) -> np.ndarray:
    """AR(1)-latent generator of tests/perf/test_tica_perf.py:65-81, vectorised with
    scipy.signal.lfilter over pre-drawn normals (draw order differs from the loop
    form, so values are NOT those of the reference's loop)."""
    from scipy.signal import lfilter

    rng = np.random.default_rng(seed)
    e = rng.normal(size=(n_frames, 3))
    latent = np.empty((n_frames, 3))
    latent[:, 0] = lfilter([1.0], [1.0, -0.985], 0.05 * e[:, 0])
    latent[:, 1] = lfilter([1.0], [1.0, -0.950], 0.08 * e[:, 1])
    latent[:, 2] = 0.5 * e[:, 2]
    mixing = rng.normal(scale=0.7, size=(3, n_features))
    noise = rng.normal(scale=0.05, size=(n_frames, n_features)).astype(np.float32)
    out = (latent @ mixing).astype(np.float32)
    out += noise
    return out


def correlated_series_loop(n_frames: int, n_features: int, seed: int) -> np.ndarray:
    """The same recipe in the reference's LOOP form (tests/perf/test_tica_perf.py:65-81): these are the values
    the reference's own benchmark feeds its estimators (the fixtures store their hash)."""
    rng = np.random.default_rng(seed)
    latent = np.zeros((n_frames, 3))
    for t in range(1, n_frames):
        latent[t, 0] = 0.985 * latent[t - 1, 0] + rng.normal(scale=0.05)
        latent[t, 1] = 0.950 * latent[t - 1, 1] + rng.normal(scale=0.08)
        latent[t, 2] = rng.normal(scale=0.5)
    mixing = rng.normal(scale=0.7, size=(3, n_features))
    noise = rng.normal(scale=0.05, size=(n_frames, n_features))
    return (latent @ mixing + noise).astype(np.float32)


def gaussian_clusters(n_clusters: int, per: int, d: int, seed: int):
    """tests/perf/test_discretize_assignment_perf.py:30-48 recipe."""
    rng = np.random.default_rng(seed)
    centers = rng.normal(loc=0.0, scale=5.0, size=(n_clusters, d))
    data = np.vstack([c + rng.normal(scale=0.2, size=(per, d)) for c in centers])
    return data.astype(np.float64), centers


def markov_labels(n: int, k: int, seed: int, stay: float = 0.9) -> np.ndarray:
    """Metastable chain on k states: stay with prob `stay`, else jump to a neighbour."""
    rng = np.random.default_rng(seed)
    jump = rng.random(n) >= stay
    step = rng.integers(1, 4, size=n) * np.where(rng.random(n) < 0.5, -1, 1)
    inc = np.where(jump, step, 0)
    inc[0] = rng.integers(0, k)
    return (np.cumsum(inc) % k).astype(np.int32)


def metastable_labels(n: int, k: int, n_macro: int, seed: int, p_leave: float = 0.01) -> np.ndarray:
    """MSM-like trajectory: `n_macro` metastable basins of k/n_macro microstates each, fast
    mixing inside a basin, rare hops to a neighbouring basin (so the spectrum has n_macro-1
    slow processes and a gap below them)."""
    rng = np.random.default_rng(seed)
    per = k // n_macro
    hop = rng.random(n) < p_leave
    step = np.where(rng.random(n) < 0.5, -1, 1)
    macro = np.cumsum(np.where(hop, step, 0))
    macro = np.abs(((macro + n_macro - 1) % (2 * n_macro - 2)) - (n_macro - 1)) if n_macro > 1 else macro * 0
    macro = np.clip(macro, 0, n_macro - 1)
    micro = rng.integers(0, per, size=n)
    return (macro * per + micro).astype(np.int32)
