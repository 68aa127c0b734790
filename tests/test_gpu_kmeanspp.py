"""k-means++ seeding on the device (pmarlo_amd/csrc/kmeanspp.hip) and the estimators built on it.

The frames drawn are a function of the data and the seed alone: oracle/npport.kmeans_plusplus (numpy, Python integers
for the 64 x 64-bit draw) must name the SAME frames.  The quality contract against the reference's estimator
(sklearn KMeans(n_init=10), S/analysis/discretize.py:458-469; deeptime KMeans('kmeans++'),
S/markov_state_model/clustering.py:322-361) tightens from 1.05 to 1.02."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import npport
from tests import _gen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,k,dtype", [(5000, 3, 17, np.float64), (40_000, 10, 100, np.float32), (3000, 45, 8, np.float64),
                                         (1025, 1, 5, np.float64), (70_000, 4, 300, np.float64)])
def test_picks_equal_the_numpy_restatement(engine, n, d, k, dtype):
    X = _gen.correlated_series(n, d, seed=n % 97).astype(dtype)
    for seed in (0, 12345, 2**40 + 7):
        cen, picked = engine.kmeans_init_plusplus(engine.to_device(X), k, seed=seed, return_picked=True)
        want_idx, want_c = npport.kmeans_plusplus(X, k, seed)
        np.testing.assert_array_equal(picked.to_host(), want_idx)
        np.testing.assert_array_equal(cen.to_host(), want_c)
        assert len(set(want_idx.tolist())) == k            # a chosen frame has weight 0 afterwards


def test_picks_with_whitening_and_degenerate_data(engine):
    rng = np.random.default_rng(3)
    X = rng.normal(size=(8000, 6)) * [1, 10, 100, 1e-3, 5, 2] + [0, 50, -3, 1, 0, 0]
    mean, std = X.mean(0), X.std(0, ddof=1)
    cen, picked = engine.kmeans_init_plusplus(engine.to_device(X), 40, seed=9, mean=engine.to_device(mean),
                                              std=engine.to_device(std), return_picked=True)
    want_idx, want_c = npport.kmeans_plusplus(X, 40, 9, mean=mean, std=std)
    np.testing.assert_array_equal(picked.to_host(), want_idx)
    np.testing.assert_array_equal(cen.to_host(), want_c)
    # fewer distinct points than centres: once every frame sits on a centre the draw is uniform (W = 0)
    Xd = np.repeat(rng.normal(size=(5, 3)), 200, axis=0)
    _, picked = engine.kmeans_init_plusplus(engine.to_device(Xd), 9, seed=1, return_picked=True)
    np.testing.assert_array_equal(picked.to_host(), npport.kmeans_plusplus(Xd, 9, 1)[0])


def test_discretizer_inertia_within_two_percent_of_the_reference_estimator(engine):
    """BASELINE config 2 (100 K x 4, k = 100): KMeansDiscretizer (k-means++ seeds, 3 restarts, full-batch Lloyd)
    against sklearn KMeans(n_clusters, random_state, n_init=10) on the whitened data."""
    from sklearn.cluster import KMeans

    from pmarlo_amd.analysis.discretize import KMeansDiscretizer

    n, d, k = 100_000, 4, 100
    X = _gen.correlated_series(n, d, seed=1000).astype(np.float64)
    mean, std = X.mean(axis=0), X.std(axis=0, ddof=1)
    Xz = (X - mean) / std
    ref = KMeans(n_clusters=k, random_state=0, n_init=10).fit(Xz)
    disc = KMeansDiscretizer(k, random_state=0)
    disc.fit(X)
    lab = disc.transform(X)
    inertia = float(((Xz - disc.centers[lab]) ** 2).sum())
    assert inertia <= float(ref.inertia_) * 1.02, (inertia, float(ref.inertia_))


def test_cluster_microstates_kmeanspp_and_minibatch(engine):
    from sklearn.cluster import KMeans

    from pmarlo_amd.markov_state_model.clustering import cluster_microstates

    n, d, k = 60_000, 5, 50
    Y = _gen.correlated_series(n, d, seed=4).astype(np.float64)
    ref = KMeans(n_clusters=k, random_state=0, n_init=10).fit(Y)

    def inertia(res):
        return float(((Y - res.centers[res.labels]) ** 2).sum())

    pp = cluster_microstates(Y, method="kmeans", n_states=k, random_state=0, n_init=3)
    assert pp.n_states == k and inertia(pp) <= float(ref.inertia_) * 1.02
    uni = cluster_microstates(Y, method="kmeans", n_states=k, random_state=0, init_strategy="uniform")
    assert uni.n_states == k
    again = cluster_microstates(Y, method="kmeans", n_states=k, random_state=0, n_init=3)
    np.testing.assert_array_equal(again.labels, pp.labels)          # same seed, same labels
    with pytest.raises(ValueError, match="init_strategy"):
        cluster_microstates(Y, method="kmeans", n_states=k, init_strategy="nearest")
    # the mini-batch estimator is real: batches are used (a coarser optimum than full-batch Lloyd, as in the reference),
    # and it is deterministic
    mb = cluster_microstates(Y, method="minibatchkmeans", n_states=k, random_state=0, batch_size=2000, max_iter=3)
    mb2 = cluster_microstates(Y, method="minibatchkmeans", n_states=k, random_state=0, batch_size=2000, max_iter=3)
    np.testing.assert_array_equal(mb.labels, mb2.labels)
    assert mb.n_states == k and inertia(mb) <= float(ref.inertia_) * 1.15
    assert inertia(mb) != inertia(pp)
    with pytest.raises(TypeError):
        cluster_microstates(Y, method="kmeans", n_states=k, batch_size=100)
