"""HIP k-means assignment vs the oracle and the sklearn-generated golden labels."""

from __future__ import annotations

import hashlib

import numpy as np
import pytest

from oracle import cport
from tests import _gen

pytestmark = pytest.mark.gpu


def _assign(engine, X, centers, mean=None, std=None, want_md=False):
    x = engine.to_device(X)
    c = engine.to_device(centers, np.float64)
    m = engine.to_device(mean, np.float64) if mean is not None else None
    s = engine.to_device(std, np.float64) if std is not None else None
    md = engine.empty((X.shape[0],), np.float64) if want_md else None
    lab = engine.kmeans_assign(x, c, mean=m, std=s, mindist=md)
    return (lab.to_host(), md.to_host()) if want_md else lab.to_host()


@pytest.mark.parametrize("prefix,xkey,lkey", [
    ("km", "km_X", "km_labels"), ("km", "km_Xq", "km_labels_q"), ("ar", "ar_X", "ar_labels")])
def test_assign_golden_sklearn_labels(engine, golden, prefix, xkey, lkey):
    g = golden("kmeans.npz")
    std = g[f"{prefix}_std"]
    got = _assign(engine, g[xkey], g[f"{prefix}_centers"], g[f"{prefix}_mean"], np.where(std > 1e-10, std, 1.0))
    np.testing.assert_array_equal(got, g[lkey])


def test_assign_golden_minibatch_branch(engine, golden):
    g = golden("kmeans.npz")
    Xb, _ = _gen.gaussian_clusters(50, 10000, 10, 99)
    if hashlib.sha256(np.ascontiguousarray(Xb).tobytes()).digest() != bytes(g["mb_input_sha"]):
        pytest.skip("numpy RNG stream differs from the one that generated the fixture")
    std = g["mb_std"]
    lab = _assign(engine, Xb, g["mb_centers"], g["mb_mean"], np.where(std > 1e-10, std, 1.0))
    np.testing.assert_array_equal(lab[:8192], g["mb_labels_head"])
    assert hashlib.sha256(lab.tobytes()).digest() == bytes(g["mb_labels_sha"])


@pytest.mark.parametrize("n,d,k,dtype", [
    (100_000, 4, 100, np.float32),    # C2
    (200_000, 10, 500, np.float64),   # C3 slice
    (50_000, 2, 20, np.float32),      # C1-like
    (20_000, 32, 300, np.float32),
    (5_000, 45, 200, np.float64),     # C4: chignolin distances, un-reduced
    (3_000, 64, 2000, np.float32),    # tiled centres
    (2_000, 256, 2000, np.float32),   # C5: clustering in the raw 256-d feature space (no TICA dims given)
    (4_000, 100, 300, np.float64),    # wide frames, d not a multiple of 4
    (3_000, 128, 700, np.float32),
    (257, 3, 5, np.float64), (1, 1, 1, np.float32)])
def test_assign_vs_oracle_bit_exact(engine, n, d, k, dtype):
    rng = np.random.default_rng(n + d + k)
    X = _gen.correlated_series(n, d, seed=d).astype(dtype) if n > 10 else rng.normal(size=(n, d)).astype(dtype)
    centers = X[rng.choice(n, size=k, replace=k > n)].astype(np.float64) + 1e-3 * rng.normal(size=(k, d))
    mean = X.mean(axis=0, dtype=np.float64)
    std = X.std(axis=0, dtype=np.float64) + 0.5
    want, md_want = cport.kmeans_assign(X.astype(np.float64), centers, mean, std, want_mindist=True)
    got, md = _assign(engine, X, centers, mean, std, want_md=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(md, md_want)
    # without whitening
    np.testing.assert_array_equal(_assign(engine, X, centers), cport.kmeans_assign(X.astype(np.float64), centers))


def test_exact_ties_go_to_lowest_index(engine):
    X = np.array([[0.0, 0.0], [1.0, 1.0], [2.0, 0.5]])
    centers = np.array([[5.0, 5.0], [1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [0.0, 0.0], [0.0, 0.0]])
    got = _assign(engine, X, centers)
    np.testing.assert_array_equal(got, cport.kmeans_assign(X, centers))
    assert got[0] == 4 and got[1] == 1


def test_assignment_is_argmin_of_true_distance(engine):
    """Property at BASELINE size (tests/perf/test_discretize_assignment_perf.py:73-151):
    the chosen centre is (one of) the nearest, for all 1M frames."""
    n, d, k = 1_000_000, 10, 500
    X = _gen.correlated_series(n, d, seed=1000)
    rng = np.random.default_rng(0)
    centers = X[rng.choice(n, k, replace=False)].astype(np.float64)
    lab, md = _assign(engine, X, centers, want_md=True)
    assert lab.min() >= 0 and lab.max() < k
    idx = rng.choice(n, 20_000, replace=False)
    Xs = X[idx].astype(np.float64)
    d2 = ((Xs[:, None, :] - centers[None, :, :]) ** 2).sum(-1)
    np.testing.assert_allclose(d2[np.arange(idx.size), lab[idx]], d2.min(axis=1), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(md[idx], d2.min(axis=1), rtol=1e-6, atol=1e-9)


def test_assign_c5_cluster_count_multi_tile(engine):
    """k = 2000 centres do not fit one LDS tile: the staged multi-tile path must agree too."""
    n, d, k = 30_000, 10, 2000
    X = _gen.correlated_series(n, d, seed=8).astype(np.float64)
    rng = np.random.default_rng(1)
    centers = X[rng.choice(n, k, replace=False)] + 1e-6 * rng.normal(size=(k, d))
    want, md_want = cport.kmeans_assign(X, centers, want_mindist=True)
    got, md = _assign(engine, X, centers, want_md=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(md, md_want)
