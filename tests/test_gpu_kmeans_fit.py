"""HIP k-means fit: bit-level agreement with the engine's CPU restatement, run-to-run
determinism, and inertia against the reference's sklearn fit (SURVEY.md hard part 2)."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import cport, npport
from tests import _gen

pytestmark = pytest.mark.gpu


def _fit(engine, X, k, seed=0, max_iter=30, tol2=0.0, mean=None, std=None):
    xd = engine.to_device(X)
    m = engine.to_device(mean, np.float64) if mean is not None else None
    s = engine.to_device(std, np.float64) if std is not None else None
    centers, state = engine.kmeans_fit(xd, k, seed=seed, max_iter=max_iter, tol2=tol2, mean=m, std=s)
    return centers.to_host(), dict(zip(engine.FIT_STATE_FIELDS, state.to_host())), xd


@pytest.mark.parametrize("n,d,k,dtype", [(20_000, 4, 100, np.float64), (50_000, 10, 50, np.float32),
                                          (5_000, 2, 8, np.float64), (3_000, 24, 40, np.float32),
                                          (3_000, 256, 150, np.float32)])  # C5 width: global-atomic member sums
def test_fit_bit_exact_vs_restated_lloyd(engine, n, d, k, dtype):
    X = _gen.correlated_series(n, d, seed=n % 13).astype(dtype)
    got, st, _ = _fit(engine, X, k, seed=7, max_iter=6)
    want, it, scale = cport.kmeans_fit(X.astype(np.float64), k, seed=7, max_iter=6)
    assert st["scale"] == scale
    np.testing.assert_array_equal(got, want)
    assert st["n_iter"] == 6 == it


def test_fit_with_whitening_matches_prewhitened(engine):
    X = _gen.correlated_series(30_000, 6, seed=5).astype(np.float64) * 7.0 + 3.0
    mean, std = X.mean(axis=0), X.std(axis=0, ddof=1)
    got, st, _ = _fit(engine, X, 64, seed=3, max_iter=5, mean=mean, std=std)
    want, _, _ = cport.kmeans_fit((X - mean) / std, 64, seed=3, max_iter=5)
    np.testing.assert_array_equal(got, want)


def test_fit_is_deterministic_and_seed_dependent(engine):
    """tests/perf/test_msm_clustering_perf.py:241-258: same seed -> same result."""
    X = _gen.correlated_series(200_000, 10, seed=1)
    a, _, _ = _fit(engine, X, 200, seed=11, max_iter=10)
    b, _, _ = _fit(engine, X, 200, seed=11, max_iter=10)
    c, _, _ = _fit(engine, X, 200, seed=12, max_iter=10)
    np.testing.assert_array_equal(a, b)
    assert not np.array_equal(a, c)


def test_fit_converges_and_stops(engine):
    X, _ = _gen.gaussian_clusters(8, 2000, 3, seed=4)
    rng = np.random.default_rng(0)
    X = X[rng.permutation(X.shape[0])]
    centers, st, xd = _fit(engine, X, 8, seed=0, max_iter=100, tol2=1e-20)
    assert st["done"] == 1.0 and st["n_iter"] < 100
    # converged centres are the member means of their own assignment
    lab = engine.kmeans_assign(xd, engine.to_device(centers)).to_host()
    for j in np.unique(lab):
        np.testing.assert_allclose(centers[j], X[lab == j].mean(axis=0), atol=1e-9)


@pytest.mark.parametrize("n,d,k", [(100_000, 4, 100), (200_000, 10, 500)])
def test_fit_inertia_vs_reference_sklearn(engine, n, d, k):
    """Contract (b): inertia no worse than the reference's estimator on the same data.
    C2-like size runs the reference's KMeans(n_init=1 here for time) branch; the
    C3-like size its MiniBatchKMeans branch (n*d >= 5e6 in the reference at 1M frames)."""
    from sklearn.cluster import KMeans, MiniBatchKMeans

    X = _gen.correlated_series(n, d, seed=1000).astype(np.float64)
    mean, std = X.mean(axis=0), X.std(axis=0, ddof=1)
    Xz = (X - mean) / std
    centers, st, xd = _fit(engine, X, k, seed=0, max_iter=40, tol2=1e-4 * d * 1e-4, mean=mean, std=std)
    md = engine.empty((n,), np.float64)
    engine.kmeans_assign(xd, engine.to_device(centers), mean=engine.to_device(mean), std=engine.to_device(std),
                         mindist=md)
    inertia = float(engine.sum_f64(md).to_host()[0])
    np.testing.assert_allclose(inertia, md.to_host().sum(), rtol=1e-12)
    if n * d >= 2_000_000:
        ref = MiniBatchKMeans(n_clusters=k, random_state=0).fit(Xz)
    else:
        ref = KMeans(n_clusters=k, random_state=0, n_init=1).fit(Xz)
    ref_inertia = float(((Xz - ref.cluster_centers_[npport.kmeans_predict(Xz, ref.cluster_centers_)]) ** 2).sum())
    assert inertia <= ref_inertia * 1.05, (inertia, ref_inertia)


def test_fit_inertia_vs_the_reference_estimator_at_c2(engine):
    """BASELINE config 2 (100 K x 4, k = 100) against the estimator the reference really runs at this size:
    sklearn KMeans(n_clusters, random_state, n_init=10) on the whitened data (S/analysis/discretize.py:458-469).
    The engine gets the same number of restarts through cluster_microstates-style seeding (10 seeded starts, lowest
    inertia) and full-batch Lloyd to sklearn's tolerance; its inertia must be within 2 % of the reference's."""
    from sklearn.cluster import KMeans

    n, d, k = 100_000, 4, 100
    X = _gen.correlated_series(n, d, seed=1000).astype(np.float64)
    mean, std = X.mean(axis=0), X.std(axis=0, ddof=1)
    Xz = (X - mean) / std
    ref = KMeans(n_clusters=k, random_state=0, n_init=10).fit(Xz)
    best = np.inf
    md = engine.empty((n,), np.float64)
    md_, sd_ = engine.to_device(mean), engine.to_device(std)
    for r in range(10):
        centers, st, xd = _fit(engine, X, k, seed=r, max_iter=300, tol2=1e-4, mean=mean, std=std)
        engine.kmeans_assign(xd, engine.to_device(centers), mean=md_, std=sd_, mindist=md)
        best = min(best, float(engine.sum_f64(md).to_host()[0]))
    assert best <= float(ref.inertia_) * 1.02, (best, float(ref.inertia_))


def test_accumulate_update_split_equals_fit(engine):
    """The shard-wise API (accumulate on two halves, summed int64, one update) equals the
    single-device fit bit for bit: what makes the multi-GPU fit shard-count independent."""
    n, d, k = 40_000, 8, 64
    X = _gen.correlated_series(n, d, seed=2).astype(np.float64)
    want, _, _ = _fit(engine, X, k, seed=5, max_iter=4)
    xa, xb = engine.to_device(X[: n // 3]), engine.to_device(X[n // 3:])
    centers, state = engine.kmeans_fit_begin(engine.to_device(X), k, seed=5, n_total=n, tol2=0.0)
    sums, counts = engine.zeros((k * d,), np.int64), engine.zeros((k,), np.int64)
    for _ in range(4):
        engine.kmeans_accumulate(xa, centers, state, sums, counts)
        engine.kmeans_accumulate(xb, centers, state, sums, counts)
        engine.kmeans_update(sums, counts, centers, state, clear=True)
    np.testing.assert_array_equal(centers.to_host(), want)


@pytest.mark.parametrize("n,d,k", [(500, 2, 4), (3000, 10, 20), (4097, 45, 7), (260, 64, 32)])
def test_silhouette_score_vs_sklearn(engine, n, d, k):
    """msm_silhouette against sklearn.metrics.silhouette_score (what _auto_select_n_states calls,
    S/markov_state_model/clustering.py:216-233); 1e-10: sklearn expands |x-y|^2 = x.x - 2x.y + y.y."""
    from sklearn.metrics import silhouette_score as sk_silhouette

    from pmarlo_amd.markov_state_model.clustering import silhouette_score

    rng = np.random.default_rng(n + k)
    X, _ = _gen.gaussian_clusters(k, n // k + 1, d, seed=k)
    X = X[:n].astype(np.float64)
    labels = rng.integers(0, k, n)
    labels[:k] = np.arange(k)
    labels[labels == k - 1] = k - 2          # one id unused, and ...
    labels[0] = k - 1                        # ... one singleton cluster (s_i = 0 by definition)
    np.testing.assert_allclose(silhouette_score(X, labels), sk_silhouette(X, labels), rtol=1e-10, atol=1e-12)


def test_auto_n_states_picks_the_planted_cluster_count(engine):
    from pmarlo_amd.markov_state_model import cluster_microstates

    X, _ = _gen.gaussian_clusters(6, 700, 3, seed=12)
    res = cluster_microstates(X, n_states="auto", random_state=3, silhouette_sample_size=2500)
    assert res.n_states == 6 and res.rationale.startswith("silhouette=") and res.rationale.endswith("sample=2500")
    assert res.labels.shape == (X.shape[0],) and res.centers.shape == (6, 3)
    res2 = cluster_microstates(X, n_states="auto", auto_n_states_override=9, random_state=3)
    assert res2.rationale == "auto-override=9" and res2.n_states == 9
    with pytest.raises(ValueError):
        cluster_microstates(X, n_states="auto", silhouette_sample_size=1)


def test_project_hands_absmax_to_fit_begin(engine):
    """msm_project's d_absmax + msm_kmeans_fit_begin(absmax_ready=1) give the same fit state (bit for bit)
    as fit_begin's own pass over Y, on the matrix-core and the wide (d > 16) projection kernels."""
    rng = np.random.default_rng(3)
    for n, F, d in ((50_000, 64, 10), (9_001, 24, 20), (777, 7, 3)):
        X = rng.normal(size=(n, F)).astype(np.float32) * 3.0
        X[5, 2] = np.nan
        xd = engine.to_device(X)
        mu = engine.to_device(np.nanmean(X.astype(np.float64), axis=0))
        inv = engine.to_device(1.0 / np.nanstd(X.astype(np.float64), axis=0))
        W = engine.to_device(rng.normal(size=(F, F)))
        state = engine.zeros((8,), np.float64)
        Y = engine.project(xd, mu, inv, W, d, absmax=state.view((1,), offset_elems=2))
        assert state.to_host()[2] == np.abs(Y.to_host()).max()
        c1, s1 = engine.kmeans_fit_begin(Y, 5, seed=1, n_total=n, tol2=0.0, state=state, absmax_ready=True)
        c2, s2 = engine.kmeans_fit_begin(Y, 5, seed=1, n_total=n, tol2=0.0)
        np.testing.assert_array_equal(s1.to_host(), s2.to_host())
        np.testing.assert_array_equal(c1.to_host(), c2.to_host())


@pytest.mark.parametrize("d,k", [(10, 300), (24, 60)])       # bf16-filter shape, all-fp64 shape
def test_incremental_member_sums_equal_full_passes(engine, d, k):
    """msm_kmeans_accumulate_delta: sums that follow only the frames that changed centre hold, pass after pass, the
    bits of a full re-accumulation (64-bit fixed-point integers), and the label buffer holds the current assignment."""
    n = 60_000
    X = _gen.correlated_series(n, d, seed=3).astype(np.float64)
    xd = engine.to_device(X)
    centers, state = engine.kmeans_fit(xd, k, seed=5, max_iter=0)
    sums_i, counts_i = engine.zeros((k * d,), np.int64), engine.zeros((k,), np.int64)
    prev = engine.empty((n,), np.int32).fill_bytes_(0xFF)
    moved = []
    for it in range(5):
        before = prev.to_host()
        engine.kmeans_accumulate(xd, centers, state, sums_i, counts_i, prev_labels=prev)
        sums_f, counts_f = engine.zeros((k * d,), np.int64), engine.zeros((k,), np.int64)
        engine.kmeans_accumulate(xd, centers, state, sums_f, counts_f)
        np.testing.assert_array_equal(sums_i.to_host(), sums_f.to_host())
        np.testing.assert_array_equal(counts_i.to_host(), counts_f.to_host())
        lab = engine.kmeans_assign(xd, centers).to_host()
        np.testing.assert_array_equal(prev.to_host(), lab)
        moved.append(int((before != lab).sum()))
        engine.kmeans_update(sums_i, counts_i, centers, state, clear=False)
    assert moved[0] == n and moved[-1] < moved[1] < n          # fewer and fewer frames move


@pytest.mark.parametrize("d,k,tol2", [(10, 300, 0.0), (10, 500, 1e-3), (6, 40, 1e-2), (24, 60, 0.0)])
def test_fused_lloyd_pass_equals_accumulate_then_update(engine, d, k, tol2):
    """msm_kmeans_lloyd_pass (the last workgroup of the accumulate launch closes the iteration; shapes outside the filter
    kernel fall back to two launches) against msm_kmeans_accumulate_delta + msm_kmeans_update: centres and the whole
    fit state (shift2, n_iter, done) bit for bit, also across the iteration where the tolerance stops the run."""
    n = 80_000
    X = _gen.correlated_series(n, d, seed=9).astype(np.float64)
    xd = engine.to_device(X)
    runs = []
    for fused in (False, True):
        centers, state = engine.kmeans_fit_begin(xd, k, seed=3, n_total=n, tol2=tol2)
        sums, counts = engine.zeros((k * d,), np.int64), engine.zeros((k,), np.int64)
        prev = engine.empty((n,), np.int32).fill_bytes_(0xFF)
        trace = []
        for _ in range(12):
            if fused:
                engine.kmeans_lloyd_pass(xd, centers, state, sums, counts, prev_labels=prev)
            else:
                engine.kmeans_accumulate(xd, centers, state, sums, counts, prev_labels=prev)
                engine.kmeans_update(sums, counts, centers, state, clear=False)
            trace.append((centers.to_host().copy(), state.to_host().copy()))
        runs.append(trace)
    for (c0, s0), (c1, s1) in zip(*runs):
        np.testing.assert_array_equal(c0, c1)
        np.testing.assert_array_equal(s0, s1)
    if tol2 > 0.0:
        assert runs[1][-1][1][5] == 1.0 or runs[1][-1][1][3] > tol2       # done flag set once the shift fell below tol2
