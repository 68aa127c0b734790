"""HIP moments / MFMA lagged covariance / Jacobi TICA solve / projection vs the oracle.

Tolerances: north_star asks TICA eigenvalues within 1e-5 relative; the kernels are
fp64 end to end, so the tests hold them to 1e-9 or tighter."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import npport
from tests import _gen

pytestmark = pytest.mark.gpu


def _bounds(segs):
    return np.asarray([a for a, _ in segs], np.int64), np.asarray([b for _, b in segs], np.int64)


@pytest.mark.parametrize("n,F,dtype,ddof", [
    (100_000, 32, np.float32, 0), (50_000, 64, np.float64, 1), (1000, 8, np.float32, 0),
    (777, 45, np.float64, 1), (3, 5, np.float32, 1), (5000, 300, np.float32, 0)])
def test_column_moments(engine, n, F, dtype, ddof):
    rng = np.random.default_rng(F)
    X = (rng.normal(size=(n, F)) * rng.uniform(0.1, 30, F) + rng.normal(size=F) * 100).astype(dtype)
    mean, std, cnt = engine.column_moments(engine.to_device(X), ddof=ddof)
    X64 = X.astype(np.float64)
    np.testing.assert_allclose(mean.to_host(), X64.mean(axis=0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(std.to_host(), X64.std(axis=0, ddof=ddof), rtol=1e-10)
    np.testing.assert_array_equal(cnt.to_host(), np.full(F, n, np.float64))


def test_column_moments_golden_preprocess_with_nans(engine, golden):
    """reduction._preprocess statistics (imputed NaNs, constant column)."""
    g = golden("tica.npz")
    X = g["pre_X"]
    mean, std, cnt = engine.column_moments(engine.to_device(X), ddof=0)
    X64 = X.astype(np.float64)
    nn = (~np.isnan(X64)).sum(axis=0)
    np.testing.assert_array_equal(cnt.to_host(), nn.astype(np.float64))
    np.testing.assert_allclose(mean.to_host(), np.nanmean(X64, axis=0), rtol=1e-13)
    # StandardScaler on the imputed matrix divides the same centred square sum by n, not by the non-NaN count
    std_imputed = std.to_host() * np.sqrt(nn / X.shape[0])
    ref = npport.preprocess(X, scale=True)
    scale = np.where(std_imputed < 10 * np.finfo(float).eps, 1.0, std_imputed)
    mine = np.where(np.isnan(X64), 0.0, (X64 - mean.to_host()) / scale)
    np.testing.assert_allclose(mine, g["pre_out_scale"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ref, g["pre_out_scale"], rtol=1e-12, atol=1e-12)


def _oracle_moments(Xc_list, lag):
    m = npport.lagged_moments(Xc_list, lag)
    return m


@pytest.mark.parametrize("n,F,lag,dtype,segs", [
    (20_000, 8, 10, np.float32, None),
    (50_000, 32, 10, np.float32, None),                         # C2 shape
    (60_000, 64, 10, np.float32, [(0, 25_000), (25_000, 25_004), (25_004, 60_000)]),  # C3 shape + short segment
    (9_000, 45, 7, np.float64, [(0, 4000), (4000, 9000)]),      # C4 feature count, fp64 input
    (5_000, 4, 1, np.float64, None),                            # C1
    (101, 16, 100, np.float32, None), (64, 64, 5, np.float32, None)])
def test_lagged_moments_vs_oracle(engine, n, F, lag, dtype, segs):
    X = _gen.correlated_series(n, F, seed=F + lag).astype(dtype)
    X64 = X.astype(np.float64)
    shift = X64.mean(axis=0)
    xd = engine.to_device(X)
    kw = {}
    seg_list = segs or [(0, n)]
    if segs:
        kw["starts"], kw["stops"] = _bounds(segs)
    mom = engine.lagged_moments(xd, lag, engine.to_device(shift), **kw).to_host()
    mom_fin = engine.lagged_moments(xd, lag, engine.to_device(shift), assume_finite=True, **kw).to_host()
    np.testing.assert_array_equal(mom, mom_fin)  # the NaN test changes nothing on finite data
    want = _oracle_moments([X64[a:b] - shift for a, b in seg_list], lag)
    M00 = mom[:F * F].reshape(F, F)
    M0t = mom[F * F:2 * F * F].reshape(F, F)
    sx, sy, T = mom[2 * F * F:2 * F * F + F], mom[2 * F * F + F:2 * F * F + 2 * F], mom[-1]
    assert T == want["T"]
    scale = np.abs(want["Mxx"]).max()
    np.testing.assert_allclose(M00, want["Mxx"], rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(M0t, want["Mxy_half"], rtol=0, atol=1e-12 * scale)
    # column sums cancel to O(1): bound the error by eps * sum|x|
    sum_abs = np.abs(X64 - shift).sum(axis=0).max()
    np.testing.assert_allclose(sx, want["sx"], rtol=0, atol=1e-13 * sum_abs)
    np.testing.assert_allclose(sy, want["sy"], rtol=0, atol=1e-13 * sum_abs)
    np.testing.assert_array_equal(M00, M00.T)
    # the symmetric flavour (S = sum (zx + zy)(zx + zy)' and M00, then (S - M00) / 2 on 32 < F <= 64): same M00 and
    # sums, the M0t block symmetrised; S carries up to 4x the magnitude, hence the wider bound
    sym = engine.lagged_moments(xd, lag, engine.to_device(shift), symmetric=True, **kw).to_host()
    sym_fin = engine.lagged_moments(xd, lag, engine.to_device(shift), symmetric=True, assume_finite=True, **kw).to_host()
    np.testing.assert_array_equal(sym, sym_fin)
    S00, S0t = sym[:F * F].reshape(F, F), sym[F * F:2 * F * F].reshape(F, F)
    np.testing.assert_allclose(S00, want["Mxx"], rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(S0t, 0.5 * (want["Mxy_half"] + want["Mxy_half"].T), rtol=0, atol=4e-12 * scale)
    np.testing.assert_array_equal(S0t, S0t.T)
    np.testing.assert_array_equal(S00, S00.T)
    np.testing.assert_allclose(sym[2 * F * F:2 * F * F + F], want["sx"], rtol=0, atol=1e-13 * sum_abs)
    np.testing.assert_allclose(sym[2 * F * F + F:2 * F * F + 2 * F], want["sy"], rtol=0, atol=1e-13 * sum_abs)
    assert sym[-1] == want["T"]


def test_lagged_moments_symmetric_exact_integers(engine):
    """Integer data: S, M00 and their difference are exact, so the symmetric flavour must reproduce (M0t + M0t') / 2 bit
    for bit on every kernel shape (F = 48 and 64: symmetric tiles; 20: plain pass + in-place symmetrisation; 80: blocked)."""
    rng = np.random.default_rng(5)
    for n, F, lag, dtype in [(1031, 64, 3, np.float32), (517, 48, 2, np.float64), (300, 20, 1, np.float32),
                             (400, 80, 4, np.float64), (333, 50, 3, np.float32), (290, 63, 1, np.float64),
                             (700, 33, 5, np.float32)]:   # 50, 63: four feature tiles without the vector loads; 33: three
        X = rng.integers(-3, 4, size=(n, F)).astype(dtype)
        X[:, 1] = np.arange(n) % 5
        segs = [(0, n // 3), (n // 3, n)]
        starts, stops = _bounds(segs)
        mom = engine.lagged_moments(engine.to_device(X), lag, engine.zeros((F,), np.float64), symmetric=True,
                                    starts=starts, stops=stops).to_host()
        want = npport.lagged_moments([X[a:b].astype(np.float64) for a, b in segs], lag)
        np.testing.assert_array_equal(mom[:F * F].reshape(F, F), want["Mxx"])
        np.testing.assert_array_equal(mom[F * F:2 * F * F].reshape(F, F), 0.5 * (want["Mxy_half"] + want["Mxy_half"].T))
        np.testing.assert_array_equal(mom[2 * F * F:2 * F * F + F], want["sx"])
        np.testing.assert_array_equal(mom[2 * F * F + F:2 * F * F + 2 * F], want["sy"])
    # lag 0 (the PCA covariance): M00 = 2 X'X, M0t = X'X
    X = rng.integers(-3, 4, size=(200, 64)).astype(np.float32)
    mom = engine.lagged_moments(engine.to_device(X), 0, engine.zeros((64,), np.float64), symmetric=True).to_host()
    G = X.astype(np.float64).T @ X.astype(np.float64)
    np.testing.assert_array_equal(mom[:64 * 64].reshape(64, 64), 2.0 * G)
    np.testing.assert_array_equal(mom[64 * 64:2 * 64 * 64].reshape(64, 64), G)


def test_lagged_moments_mfma_layout_asymmetric(engine):
    """Exact-integer data with an asymmetric lag structure catches any row/col swap
    in the fp64 MFMA C/D layout (results are exact integers)."""
    n, F, lag = 257, 48, 3
    rng = np.random.default_rng(2)
    X = rng.integers(-3, 4, size=(n, F)).astype(np.float64)
    X[:, 1] = np.arange(n) % 5
    mom = engine.lagged_moments(engine.to_device(X), lag, engine.zeros((F,), np.float64)).to_host()
    want = npport.lagged_moments([X], lag)
    np.testing.assert_array_equal(mom[:F * F].reshape(F, F), want["Mxx"])
    np.testing.assert_array_equal(mom[F * F:2 * F * F].reshape(F, F), want["Mxy_half"])
    assert not np.array_equal(want["Mxy_half"], want["Mxy_half"].T)


def test_lagged_moments_nan_is_imputed_to_mean(engine):
    n, F, lag = 4000, 16, 5
    X = _gen.correlated_series(n, F, 4).astype(np.float64)
    Xn = X.copy()
    Xn[17, 3] = np.nan
    Xn[2000, 0] = np.nan
    shift = np.nanmean(Xn, axis=0)
    mom = engine.lagged_moments(engine.to_device(Xn), lag, engine.to_device(shift)).to_host()
    Xi = np.where(np.isnan(Xn), shift, Xn) - shift
    want = npport.lagged_moments([Xi], lag)
    np.testing.assert_allclose(mom[:F * F].reshape(F, F), want["Mxx"], atol=1e-11)


@pytest.mark.parametrize("n", [1, 2, 3, 8, 31, 64, 65, 100, 200])
def test_jacobi_eigh(engine, n):
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, n))
    A = B @ B.T / n + np.diag(rng.uniform(0, 2, n))
    w, v, sweeps = engine.eigh(engine.to_device(A))
    w, v = w.to_host(), v.to_host()
    w_ref = np.linalg.eigvalsh(A)
    np.testing.assert_allclose(w, w_ref, rtol=1e-12, atol=1e-13 * np.abs(w_ref).max())
    np.testing.assert_allclose(v.T @ v, np.eye(n), atol=1e-12)
    np.testing.assert_allclose(A @ v, v * w[None, :], atol=1e-11 * np.abs(w_ref).max())
    assert int(sweeps.to_host()[0]) <= 14


def test_jacobi_eigh_degenerate_and_indefinite(engine):
    A = np.diag([3.0, 3.0, -1.0, 0.0, 3.0])
    A[0, 1] = A[1, 0] = 0.0
    w = engine.eigh(engine.to_device(A))[0].to_host()
    np.testing.assert_allclose(w, [-1, 0, 3, 3, 3], atol=1e-14)


def _tica_gpu(engine, X, lag, dim, segs=None, scale=True):
    n, F = X.shape
    xd = engine.to_device(X)
    mean, std, cnt = engine.column_moments(xd, ddof=0)
    std_h = std.to_host()
    sc = np.where(std_h < 10 * np.finfo(float).eps, 1.0, std_h) if scale else np.ones(F)
    scale_d = engine.to_device(sc)
    kw = {}
    if segs:
        kw["starts"], kw["stops"] = _bounds(segs)
    mom = engine.lagged_moments(xd, lag, mean, **kw)
    eig, W, m2, rank = engine.tica_solve(mom, F, scale=scale_d)
    Y = engine.project(xd, mean, engine.to_device(1.0 / sc), W, dim, mean2=m2)
    return eig.to_host(), W.to_host(), m2.to_host(), int(rank.to_host()[0]), Y.to_host()


@pytest.mark.parametrize("n,F,lag,dim", [(4000, 8, 10, 4), (100_000, 32, 10, 4), (50_000, 64, 10, 10),
                                          (3000, 45, 5, 3), (2000, 2, 1, 2)])
def test_tica_pipeline_vs_oracle(engine, n, F, lag, dim):
    X = _gen.correlated_series(n, F, seed=1000 + F)
    eig, W, m2, rank, Y = _tica_gpu(engine, X, lag, dim)
    Xp = npport.preprocess(X, scale=True)
    model = npport.tica_fit([Xp], lag, dim=dim)
    assert rank == model["rank"]
    # eigenvalues: north_star tolerance 1e-5 relative; fp64 kernels reach ~1e-10
    np.testing.assert_allclose(eig[:rank], model["eigenvalues"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m2, model["mean"], atol=1e-12)
    Yo = npport.tica_transform(model, Xp)
    # columns agree up to sign (reference test tolerance: atol 1e-6, test_reduction.py:43-60)
    for c in range(dim):
        s = np.sign(np.dot(Y[:, c], Yo[:, c])) or 1.0
        np.testing.assert_allclose(s * Y[:, c], Yo[:, c], atol=1e-8 * max(1.0, np.abs(Yo[:, c]).max()))
    # canonical signs should in fact agree on generic data
    signs = [np.sign(np.dot(W[:, c], model["coefficients"][:, c])) for c in range(dim)]
    assert all(s > 0 for s in signs)


def test_tica_golden_eigenvalues_against_reference_estimator(engine, golden):
    """The reference's own in-repo TICA eigenvalue estimator (a9) on its benchmark input:
    the reversible estimator agrees to O(lag/N) with it, and the slow modes sit at the
    analytic AR(1) values."""
    g = golden("tica.npz")
    eig, *_ = _tica_gpu(engine, g["tica_X"], int(g["tica_lag"]), 4)
    np.testing.assert_allclose(eig[:2], g["tica_top_eigs"][:2], atol=2e-2)
    assert abs(eig[0] - 0.985 ** 10) < 0.08 and abs(eig[1] - 0.95 ** 10) < 0.1


def test_tica_rank_deficient_input(engine):
    """Duplicate / constant columns: epsilon cut-off must drop them (deeptime spd_inv_split)."""
    X = _gen.correlated_series(5000, 6, 9).astype(np.float64)
    X = np.hstack([X, X[:, :2] * 2.0 + 1.0, np.full((5000, 1), 3.0)])
    eig, W, m2, rank, Y = _tica_gpu(engine, X, 4, 3)
    model = npport.tica_fit([npport.preprocess(X, scale=True)], 4, dim=3)
    assert rank == model["rank"] == 6
    np.testing.assert_allclose(eig[:rank], model["eigenvalues"], rtol=1e-7, atol=1e-9)
    assert np.all(eig[rank:] == 0) and np.all(W[:, rank:] == 0)


@pytest.mark.parametrize("var_small,rank_want", [(1.5e-6, 8), (0.6e-6, 6)])
def test_tica_directions_near_the_epsilon_cut(engine, var_small, rank_want):
    """Two directions whose variance sits just above / below deeptime's epsilon = 1e-6.  Above: the cheap certificate
    lambda_min >= 1 / ||W||_F^2 is too coarse (2 / 1.5e-6 > 1e6), so the elimination of the probe C00 - epsilon I must
    run and keep the full rank; below: the probe fails and the eigen path cuts the two directions."""
    rng = np.random.default_rng(3)
    X = _gen.correlated_series(20_000, 6, seed=17).astype(np.float64)
    X = X / X.std(axis=0)
    X = np.hstack([X, rng.normal(0.0, np.sqrt(var_small), size=(20_000, 2))])
    eig, W, m2, rank, Y = _tica_gpu(engine, X, 5, 3, scale=False)
    model = npport.tica_fit([npport.preprocess(X, scale=False)], 5, dim=3)
    assert rank == model["rank"] == rank_want
    np.testing.assert_allclose(eig[:rank], model["eigenvalues"], rtol=1e-7, atol=1e-9)


def test_tica_multi_trajectory_segments(engine):
    """_maybe_apply_tica fits on a list of trajectories: pairs never cross a boundary."""
    parts = [_gen.correlated_series(m, 12, seed=s) for m, s in [(3000, 1), (50, 2), (4000, 3), (8, 4)]]
    X = np.vstack(parts)
    edges = np.cumsum([0] + [p.shape[0] for p in parts])
    segs = [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:])]
    eig, W, m2, rank, Y = _tica_gpu(engine, X, 10, 5, segs=segs)
    Xp = npport.preprocess(X, scale=True)
    model = npport.tica_fit([Xp[a:b] for a, b in segs], 10, dim=5)
    np.testing.assert_allclose(eig[:rank], model["eigenvalues"], rtol=1e-9, atol=1e-11)
    assert model["T"] == (3000 - 10) + (50 - 10) + (4000 - 10)


def test_project_matches_numpy(engine):
    rng = np.random.default_rng(0)
    n, F, d = 10_000, 64, 10
    X = rng.normal(size=(n, F)).astype(np.float32)
    mu, isg, m2 = rng.normal(size=F), rng.uniform(0.5, 2, F), rng.normal(size=F) * 0.01
    W = rng.normal(size=(F, F))
    Y = engine.project(engine.to_device(X), engine.to_device(mu), engine.to_device(isg), engine.to_device(W), d,
                       mean2=engine.to_device(m2)).to_host()
    want = ((X.astype(np.float64) - mu) * isg - m2) @ W[:, :d]
    np.testing.assert_allclose(Y, want, rtol=1e-12, atol=1e-12)
    # the entry for NaN-free input leaves out a test, not arithmetic: the same bits; shapes off the vector path too
    for dtype, Fx in [(np.float32, 64), (np.float64, 64), (np.float32, 48), (np.float32, 50)]:
        Xd = engine.to_device(X[:, :Fx].astype(dtype))
        args = (engine.to_device(mu[:Fx]), engine.to_device(isg[:Fx]), engine.to_device(np.ascontiguousarray(W[:Fx, :Fx])), d)
        a = engine.project(Xd, *args, mean2=engine.to_device(m2[:Fx])).to_host()
        b = engine.project(Xd, *args, mean2=engine.to_device(m2[:Fx]), assume_finite=True).to_host()
        np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(a, ((X[:, :Fx].astype(dtype).astype(np.float64) - mu[:Fx]) * isg[:Fx] - m2[:Fx]) @ W[:Fx, :d],
                                   rtol=1e-12, atol=1e-12)
    # NaN -> column mean (z = 0) on the plain entry
    Xn = X.copy()
    Xn[5, 3] = np.nan
    Yn = engine.project(engine.to_device(Xn), engine.to_device(mu), engine.to_device(isg), engine.to_device(W), d,
                        mean2=engine.to_device(m2)).to_host()
    Xi = Xn.astype(np.float64)
    Xi[5, 3] = mu[3]
    np.testing.assert_allclose(Yn, ((Xi - mu) * isg - m2) @ W[:, :d], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n,F,lag,dtype,segs", [
    (6_000, 128, 5, np.float32, None),
    (5_000, 256, 10, np.float32, [(0, 1777), (1777, 5000)]),      # C5 feature count
    (3_000, 100, 3, np.float64, None),                            # not a multiple of 64: guarded path
    (2_000, 65, 2, np.float32, None)])
def test_lagged_moments_blocked_large_F(engine, n, F, lag, dtype, segs):
    X = _gen.correlated_series(n, F, seed=F).astype(dtype)
    X64 = X.astype(np.float64)
    shift = X64.mean(axis=0)
    kw = {}
    seg_list = segs or [(0, n)]
    if segs:
        kw["starts"], kw["stops"] = _bounds(segs)
    mom = engine.lagged_moments(engine.to_device(X), lag, engine.to_device(shift), **kw).to_host()
    want = npport.lagged_moments([X64[a:b] - shift for a, b in seg_list], lag)
    M00 = mom[:F * F].reshape(F, F)
    M0t = mom[F * F:2 * F * F].reshape(F, F)
    scale = np.abs(want["Mxx"]).max()
    np.testing.assert_allclose(M00, want["Mxx"], rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(M0t, want["Mxy_half"], rtol=0, atol=1e-12 * scale)
    sum_abs = np.abs(X64 - shift).sum(axis=0).max()
    np.testing.assert_allclose(mom[2 * F * F:2 * F * F + F], want["sx"], rtol=0, atol=1e-13 * sum_abs)
    np.testing.assert_allclose(mom[2 * F * F + F:2 * F * F + 2 * F], want["sy"], rtol=0, atol=1e-13 * sum_abs)
    assert mom[-1] == want["T"]
    np.testing.assert_array_equal(M00, M00.T)


def test_tica_pipeline_c5_feature_count(engine):
    """C5 shape at reduced N: F = 256 features -> TICA (block-task covariance, global-memory
    Jacobi) -> projection, against the restated deeptime estimator."""
    n, F, lag, dim = 20_000, 256, 10, 10
    X = _gen.correlated_series(n, F, seed=5)
    eig, W, m2, rank, Y = _tica_gpu(engine, X, lag, dim)
    Xp = npport.preprocess(X, scale=True)
    model = npport.tica_fit([Xp], lag, dim=dim)
    assert rank == model["rank"] == F
    np.testing.assert_allclose(eig[:dim], model["eigenvalues"][:dim], rtol=1e-8, atol=1e-10)
    Yo = npport.tica_transform(model, Xp)
    for c in range(3):  # the resolved slow modes; the noise modes are near-degenerate at this N
        s = np.sign(np.dot(Y[:, c], Yo[:, c])) or 1.0
        np.testing.assert_allclose(s * Y[:, c], Yo[:, c], atol=1e-7 * max(1.0, np.abs(Yo[:, c]).max()))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_moments_from_lagged_equals_the_separate_pass(engine, dtype):
    """msm_moments_from_lagged: the standardisation sums recovered from the lagged moments plus the edge
    frames equal msm_column_moments_partial's (same shift) to rounding, for several segments including
    ones shorter than the lag (no pairs at all) and exactly lag + 1 frames."""
    rng = np.random.default_rng(11)
    n, F, lag = 40_000, 24, 7
    X = (rng.normal(size=(n, F)) * rng.uniform(0.2, 5.0, size=F) + rng.normal(size=F) * 10).astype(dtype)
    xd = engine.to_device(X)
    for segs in (None, [(0, 15_000), (15_000, 15_005), (15_010, 15_018), (20_000, 40_000)], [(3, 3 + lag)]):
        starts = stops = None
        if segs is not None:
            starts = np.asarray([a for a, _ in segs], np.int64)
            stops = np.asarray([b for _, b in segs], np.int64)
        rows = np.concatenate([X[a:b] for a, b in (segs or [(0, n)])]).astype(np.float64)
        shift = engine.to_device(rows[0].copy())
        mom = engine.lagged_moments(xd, lag, shift, starts=starts, stops=stops, assume_finite=True)
        got = engine.moments_from_lagged(xd, lag, shift, mom, starts=starts, stops=stops).to_host()
        z = rows - rows[0]
        np.testing.assert_array_equal(got[:F], float(rows.shape[0]))
        np.testing.assert_allclose(got[F:2 * F], z.sum(0), rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(got[2 * F:], (z * z).sum(0), rtol=1e-12)


# ---- the reference's own in-repo TICA eigenvalue estimator, on the device -----------------
def test_top_eigenvalues_match_reference_golden(engine, golden):
    """_estimate_top_eigenvalues (S/features/deeptica/core/trainer_api.py:632-656) is the one TICA function of the
    reference that runs in the build container: its outputs (fixtures made by importing it) pin the device path --
    one-sided fp64 matrix-core moments + the on-device eigensolves -- to <= 1e-9."""
    from pmarlo_amd.features.deeptica.core import trainer_api

    g = golden("tica.npz")
    Yp = npport.preprocess(g["tica_X"], scale=True)
    lag = int(g["tica_lag"])
    idx = np.arange(Yp.shape[0] - lag)
    ev = trainer_api.estimate_top_eigenvalues(Yp, idx, idx + lag, 8, engine=engine)
    np.testing.assert_allclose(ev, g["tica_top_eigs"], rtol=1e-9, atol=1e-12)
    assert trainer_api._estimate_top_eigenvalues(Yp, idx[:0], idx[:0], None) is None


def test_top_eigenvalues_on_the_reference_benchmark_input(engine, golden):
    """The reference's benchmark of this estimator (tests/perf/test_tica_perf.py:216-235): raw float32 AR(1) series,
    N = 20 000, F = 8, seed 21, lag 10; then arbitrary (non-run) index pairs, which take the gathered path."""
    import hashlib
    from types import SimpleNamespace

    from pmarlo_amd.features.deeptica.core import trainer_api

    g = golden("tica.npz")
    P = _gen.correlated_series_loop(20_000, 8, 21)
    if hashlib.sha256(np.ascontiguousarray(P).tobytes()).digest() != bytes(g["perf_input_sha"]):
        pytest.skip("numpy RNG stream differs from the one that generated the fixture")
    lag = int(g["perf_lag"])
    idx = np.arange(P.shape[0] - lag)
    it = np.arange(0, P.shape[0] - 20, 3)
    itau = it + np.where(np.arange(it.size) % 2 == 0, 7, 13)
    # the series as float64: the reference's fp64 answer, held to 1e-9
    P64 = P.astype(np.float64)
    ev = trainer_api._estimate_top_eigenvalues(P64, idx, idx + lag, SimpleNamespace(n_out=8))
    assert isinstance(ev, list) and len(ev) == 8
    np.testing.assert_allclose(ev, g["perf_top_eigs_f64"], rtol=1e-9, atol=1e-12)
    assert len(trainer_api._estimate_top_eigenvalues(P64, idx, idx + lag, SimpleNamespace())) == 2   # default n_out
    ev2 = trainer_api.estimate_top_eigenvalues(P64, it, itau, 8, engine=engine)
    np.testing.assert_allclose(ev2, g["perf_pairs_eigs_f64"], rtol=1e-9, atol=1e-12)
    # the float32 series as the benchmark passes it: the reference then computes in float32 (its answer is 1e-6
    # away from its own fp64 one); the device path reads the float32 frames and accumulates in fp64, so it lands
    # on the fp64 values -- within north_star's 1e-5 of what the reference returns
    ev32 = trainer_api.estimate_top_eigenvalues(P, idx, idx + lag, 8, engine=engine)
    np.testing.assert_allclose(ev32, g["perf_top_eigs_f64"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ev32[:2], g["perf_top_eigs"][:2], rtol=1e-5)
    np.testing.assert_allclose(ev32, g["perf_top_eigs"], atol=5e-6)


def test_top_eigenvalues_pair_runs_and_shapes(engine):
    """Pairs listed shard by shard (40 runs: more than one moments call), lag 0, and F = 64 / 100 against the numpy
    restatement of the estimator."""
    from pmarlo_amd.features.deeptica.core import trainer_api

    X = _gen.correlated_series(40_000, 16, 5).astype(np.float64)
    lag = 7
    idx = np.concatenate([np.arange(s, s + 1000 - lag) for s in range(0, 40_000, 1000)])
    got = trainer_api.estimate_top_eigenvalues(X, idx, idx + lag, 16, engine=engine)
    np.testing.assert_allclose(got, npport.estimate_top_eigenvalues(X, idx, idx + lag, 16), rtol=1e-9, atol=1e-12)
    i0 = np.arange(500, 900)
    np.testing.assert_allclose(trainer_api.estimate_top_eigenvalues(X, i0, i0, 16, engine=engine),
                               npport.estimate_top_eigenvalues(X, i0, i0, 16), rtol=1e-9, atol=1e-12)
    for F in (64, 100):
        Z = _gen.correlated_series(30_000, F, F)
        i1 = np.arange(30_000 - 10)
        np.testing.assert_allclose(trainer_api.estimate_top_eigenvalues(Z, i1, i1 + 10, 10, engine=engine),
                                   npport.estimate_top_eigenvalues(Z.astype(np.float64), i1, i1 + 10, 10), rtol=1e-8,
                                   atol=1e-11)
