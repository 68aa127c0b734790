"""The certified bf16 filter of the k-means passes (pmarlo_amd/csrc/kmeans_filter.h) must give the labels,
ties, distances and member sums of the pinned fp64 arithmetic bit for bit: adversarial inputs against the C
oracle (oracle/msm_oracle.c: the fma chain of msm_kmeans_assign)."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import cport
from tests import _gen

pytestmark = pytest.mark.gpu


def _assign(engine, X, centers, mean=None, std=None, image=False):
    x = engine.to_device(X)
    c = engine.to_device(np.ascontiguousarray(centers, np.float64))
    m = engine.to_device(np.asarray(mean, np.float64)) if mean is not None else None
    s = engine.to_device(np.asarray(std, np.float64)) if std is not None else None
    md = engine.empty((X.shape[0],), np.float64)
    img = engine.kmeans_pack(x, mean=m, std=s) if image else None
    lab = engine.kmeans_assign(x, c, mean=m, std=s, mindist=md, image=img)
    return lab.to_host(), md.to_host()


def _check(engine, X, centers, mean=None, std=None):
    want, md_want = cport.kmeans_assign(np.asarray(X, np.float64), np.asarray(centers, np.float64), mean, std,
                                        want_mindist=True)
    for image in (False, True):
        got, md = _assign(engine, X, centers, mean, std, image=image)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(md, md_want)
    return want


@pytest.mark.parametrize("d", list(range(1, 11)))
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_every_filter_width(engine, d, dtype):
    """d = 1..4 run one matrix instruction per tile, 5..10 two; odd tile counts, padding centres."""
    rng = np.random.default_rng(100 + d)
    n, k = 30_011, [7, 16, 33, 100, 250, 500, 17, 480, 512, 750][d - 1]
    X = _gen.correlated_series(n, d, seed=d).astype(dtype)
    centers = X[rng.choice(n, size=k, replace=False)].astype(np.float64) + 1e-4 * rng.normal(size=(k, d))
    engine.kmeans_filter_scanned(reset=True)
    _check(engine, X, centers)
    scanned = engine.kmeans_filter_scanned()
    assert scanned < 0.2 * 2 * n, f"the filter certified too few frames: {scanned} scans for {2 * n} frames"
    _check(engine, X, centers, X.mean(0, dtype=np.float64), X.std(0, dtype=np.float64) + 0.25)


@pytest.mark.parametrize("d", [1, 2, 4])
@pytest.mark.parametrize("k", [1025, 1100, 1400])
def test_more_than_64_tiles(engine, d, k):
    """One matrix instruction per tile (d <= 4) leaves LDS room for more than 64 tiles of centres: assign (with and
    without a prebuilt image), the whole fit and the delta-mode passes, against the C restatement."""
    rng = np.random.default_rng(1000 * d + k)
    n = 40_009
    X = np.cumsum(rng.normal(size=(n, d)), axis=0) * 0.05 + rng.normal(size=(n, d))
    centers = X[rng.choice(n, size=k, replace=False)] + 1e-5 * rng.normal(size=(k, d))
    _check(engine, X, centers)
    want_c, _, _ = cport.kmeans_fit(X, k, seed=5, max_iter=4, tol2=0.0)
    x = engine.to_device(X)
    got_c, _ = engine.kmeans_fit(x, k, seed=5, max_iter=4, tol2=0.0)
    np.testing.assert_array_equal(got_c.to_host(), want_c)
    # pass by pass: full sums against delta sums on one image
    img = engine.kmeans_pack(x)
    c2, st2 = engine.kmeans_fit_begin(x, k, seed=5, n_total=n, tol2=0.0)
    c3, st3 = engine.kmeans_fit_begin(x, k, seed=5, n_total=n, tol2=0.0)
    sums2, counts2 = engine.zeros((k * d,), np.int64), engine.zeros((k,), np.int64)
    sums3, counts3 = engine.zeros((k * d,), np.int64), engine.zeros((k,), np.int64)
    prev = engine.empty((n,), np.int32)
    prev.fill_bytes_(0xFF)
    for _ in range(4):
        engine.kmeans_accumulate(x, c2, st2, sums2, counts2, image=img)
        engine.kmeans_update(sums2, counts2, c2, st2, clear=True)
        engine.kmeans_accumulate(x, c3, st3, sums3, counts3, image=img, prev_labels=prev)
        engine.kmeans_update(sums3, counts3, c3, st3, clear=False)
    np.testing.assert_array_equal(c2.to_host(), want_c)
    np.testing.assert_array_equal(c3.to_host(), want_c)


def test_near_ties_and_duplicates(engine):
    """Frames on bisector planes, centres that differ in the last bits, duplicated centres: every frame the
    filter cannot certify must come out of the exhaustive scan with the pinned tie rule (lowest index)."""
    rng = np.random.default_rng(5)
    d, k = 10, 500
    base = rng.normal(size=(k // 2, d))
    eps = np.ldexp(1.0, -rng.integers(20, 52, size=(k // 2, 1)))
    centers = np.vstack([base, base * (1.0 + eps * rng.choice([-1, 0, 1], size=(k // 2, d)))])
    centers[7] = centers[300]              # exact duplicates in different tiles / lanes
    centers[301] = centers[8]
    centers[499] = centers[0]
    perm = rng.permutation(k)
    centers = centers[perm]
    # frames: exact midpoints of centre pairs (ties up to round-off), centres themselves, random points
    a, b = rng.integers(0, k, size=(2, 20_000))
    X = np.vstack([0.5 * (centers[a] + centers[b]), centers, centers + 1e-9 * rng.normal(size=(k, d)),
                   rng.normal(size=(20_000, d))])
    engine.kmeans_filter_scanned(reset=True)
    _check(engine, X, centers)
    assert engine.kmeans_filter_scanned() > 0          # frames sitting on duplicated centres cannot be certified
    # two copies of one centre, nothing else near: the lower index wins
    c = np.array([[3.0, 1.0], [0.0, 0.0], [3.0, 1.0], [0.0, 0.0], [9.0, 9.0]] + [[50.0 + i, -7.0] for i in range(40)])
    got, _ = _assign(engine, np.array([[3.0, 1.0], [0.1, 0.0], [2.9, 1.2]]), c)
    np.testing.assert_array_equal(got, [0, 1, 0])


def test_out_of_range_and_non_finite_frames(engine):
    """NaN, inf, huge, tiny and zero coordinates.  Tiny ones (0 < |v| < 1e-14) stay out of the bf16 images and are paid
    for in the bound; huge / non-finite ones send the frame (for a centre: every frame) to the exhaustive scan; labels stay
    those of the fp64 arithmetic."""
    rng = np.random.default_rng(9)
    n, d, k = 5000, 6, 64
    X = rng.normal(size=(n, d))
    X[10, 2] = np.nan
    X[11] = np.nan
    X[12, 0] = np.inf
    X[13, 5] = -np.inf
    X[14] = 1e200
    X[15, 1] = 1e19
    X[16] = 1e-300
    X[17, 3] = 3e-15
    X[18] = 0.0
    X[19, 4] = 0.0
    centers = rng.normal(size=(k, d))
    _check(engine, X, centers)
    c2 = centers.copy()
    c2[5, 1] = 1e-20                       # tiny centre coordinates: left out of the filter, covered by its bound
    c2[6, 0] = -3e-15
    c2[9] = 0.0
    engine.kmeans_filter_scanned(reset=True)
    _check(engine, X, c2)
    assert engine.kmeans_filter_scanned() < 0.2 * 2 * n
    Xt = X.copy()
    Xt[:, 3] *= 1e-17                      # a whole feature at round-off level (a centred constant column)
    c2[:, 3] *= 1e-17
    engine.kmeans_filter_scanned(reset=True)
    _check(engine, Xt, c2)
    assert engine.kmeans_filter_scanned() < 0.2 * 2 * n
    c4 = centers.copy()
    c4[5, 1] = 1e19                        # a centre outside the range: everything is scanned
    engine.kmeans_filter_scanned(reset=True)
    _check(engine, X, c4)
    assert engine.kmeans_filter_scanned() >= 2 * (n - 20)
    c4[5, 1] = np.nan
    _check(engine, X, c4)
    c3 = centers * 1e17                    # inside the range, products near the top of fp32
    _check(engine, X * 1e17, c3)
    _check(engine, X * 1e-13, centers * 1e-13)
    _check(engine, X * 1e5, centers * 1e-9)


def test_scores_with_heavy_cancellation(engine):
    """Data far from the origin relative to its spread: x.c and |c|^2/2 cancel to 1e-7 of their size, the regime
    in which the filter's bound is widest relative to the score gaps."""
    rng = np.random.default_rng(21)
    n, d, k = 40_000, 10, 400
    X = 1000.0 + rng.normal(size=(n, d)) * 0.01
    centers = X[rng.choice(n, size=k, replace=False)] + 1e-5 * rng.normal(size=(k, d))
    _check(engine, X, centers)
    _check(engine, X.astype(np.float32), centers)


def test_fit_member_sums_bit_exact(engine):
    """assign + accumulate through the filter: the int64 member sums / counts of the engine's Lloyd fit equal the
    C restatement's, with and without a prebuilt image, and do not depend on the image being rebuilt."""
    n, d, k = 60_000, 10, 500
    X = _gen.correlated_series(n, d, seed=3).astype(np.float64)
    want_c, _, _ = cport.kmeans_fit(X, k, seed=11, max_iter=6, tol2=0.0)
    x = engine.to_device(X)
    centers, state = engine.kmeans_fit(x, k, seed=11, max_iter=6, tol2=0.0)
    np.testing.assert_array_equal(centers.to_host(), want_c)
    # the same fit driven pass by pass with one image
    img = engine.kmeans_pack(x)
    c2, st2 = engine.kmeans_fit_begin(x, k, seed=11, n_total=n, tol2=0.0)
    sums = engine.zeros((k * d,), np.int64)
    counts = engine.zeros((k,), np.int64)
    for _ in range(6):
        engine.kmeans_accumulate(x, c2, st2, sums, counts, image=img)
        engine.kmeans_update(sums, counts, c2, st2, clear=True)
    np.testing.assert_array_equal(c2.to_host(), want_c)


def test_large_shard_labels(engine):
    """1 M frames (the bench shard shape, d = 10, k = 500): oracle labels on a sample, plus the size-independent
    property that every label is the arg-min of the true distance."""
    n, d, k = 1_000_000, 10, 500
    rng = np.random.default_rng(77)
    Y = _gen.correlated_series(n, d, seed=5).astype(np.float64)
    centers = Y[rng.choice(n, size=k, replace=False)] + 1e-6 * rng.normal(size=(k, d))
    x = engine.to_device(Y)
    engine.kmeans_filter_scanned(reset=True)
    lab = engine.kmeans_assign(x, engine.to_device(centers)).to_host()
    scanned = engine.kmeans_filter_scanned()
    assert scanned < 0.1 * n
    idx = rng.choice(n, size=100_000, replace=False)
    np.testing.assert_array_equal(lab[idx], cport.kmeans_assign(Y[idx], centers))
    # property on all frames: the chosen centre is within round-off of the nearest one
    d_own = np.einsum("ij,ij->i", Y - centers[lab], Y - centers[lab])
    for lo in range(0, n, 200_000):
        blk = Y[lo:lo + 200_000]
        dist = (blk ** 2).sum(1)[:, None] - 2.0 * blk @ centers.T + (centers ** 2).sum(1)[None, :]
        assert np.all(d_own[lo:lo + 200_000] <= dist.min(1) + 1e-9)
