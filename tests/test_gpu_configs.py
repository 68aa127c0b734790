"""End-to-end parity at the BASELINE.json configurations (reduced frame counts where the CPU
oracle would take minutes): the whole device chain against the whole oracle chain, stage by stage.

Bars (north star): features 2e-6 relative (fp32 featurizer), TICA eigenvalues 1e-9 relative,
labels bit-exact GIVEN the same projected coordinates and centres (the oracle re-assigns the
device's Y with the device's centres: any differing label is a kernel bug, not a k-means RNG
effect), transition counts bit-exact, implied timescales 1e-6 relative."""
import numpy as np
import pytest

from oracle import cport, npport
from pmarlo_amd.pipeline import MSMPipeline
from tests import _gen

pytestmark = pytest.mark.gpu


def _tiled_frames(base_xyz, n_frames, seed, sigma=0.02):
    """tests/perf/test_feature_featurization_perf.py:35-47 recipe: tile the models, add N(0, sigma nm)."""
    rng = np.random.default_rng(seed)
    reps = -(-n_frames // base_xyz.shape[0])
    xyz = np.tile(base_xyz, (reps, 1, 1))[:n_frames].astype(np.float32)
    return xyz + rng.normal(0.0, sigma, size=xyz.shape).astype(np.float32)


def _slow_walk(n_frames, n_atoms, seed, amp=0.08):
    """A slowly varying collective displacement so the molecular configs have kinetics to find."""
    rng = np.random.default_rng(seed)
    modes = rng.normal(size=(2, n_atoms, 3)).astype(np.float32)
    z = np.zeros((n_frames, 2))
    e = rng.normal(size=(n_frames, 2))
    for t in range(1, n_frames):
        z[t] = 0.98 * z[t - 1] + 0.2 * e[t]
    return (amp * np.tensordot(z, modes, axes=(1, 0))).astype(np.float32)


def test_c1_alanine_phi_psi_k20_lag1(engine, golden):
    """C1: alanine dipeptide, phi/psi features (+ cos/sin), 20 microstates, lag 1."""
    g = golden("featurizer.npz")
    n = 20_000
    xyz = _tiled_frames(g["ala_xyz"][:1], n, seed=1234) + _slow_walk(n, 22, seed=5)
    quads = g["ala_quads"]
    X = engine.featurize(engine.to_device(xyz), quads=quads, dihedral_mode=1)          # [cos, sin] per angle
    Xh = X.to_host()
    ang = npport.dihedrals(xyz.astype(np.float64), quads)
    want, _ = npport.trig_expand_periodic(ang, np.ones(2, dtype=bool))
    np.testing.assert_allclose(Xh, want, rtol=0, atol=3e-5)
    pipe = MSMPipeline(engine)
    labels, centers, _ = pipe.cluster(X, 20, seed=3, max_iter=30)
    lab = labels.to_host()
    np.testing.assert_array_equal(lab, cport.kmeans_assign(Xh.astype(np.float64), centers.to_host()))
    counts, pairs = pipe.count(labels, 20, 1)
    C, p = cport.count_transitions(lab, 20, 1)
    np.testing.assert_array_equal(counts.to_host(), C)
    assert int(pairs.to_host()[0]) == p == n - 1
    est = pipe.estimate(counts, n_its=3, lag=1.0)
    _, ts_ref = npport.its_from_counts(C.astype(float), 1, 3)
    np.testing.assert_allclose(est["spectrum"]["its_ts"][0], ts_ref, rtol=1e-6)


@pytest.mark.parametrize("n,F,d,k,lag,segments", [
    (100_000, 32, 4, 100, 10, None),                                   # C2 as specified
    (120_000, 64, 10, 500, 10, [(0, 30_000), (30_000, 120_000)]),      # C3 shape at reduced N, two trajectories
])
def test_c2_c3_synthetic_tica_kmeans_counts_its(engine, n, F, d, k, lag, segments):
    X = _gen.correlated_series(n, F, seed=1000)
    pipe = MSMPipeline(engine)
    res = pipe.run(engine.to_device(X), lag=lag, tica_dim=d, k=k, segments=segments, seed=0, kmeans_iter=10)
    parts = [X] if segments is None else [X[a:b] for a, b in segments]
    Xp = npport.preprocess(X, scale=True)
    ref = npport.tica_fit([Xp] if segments is None else [Xp[a:b] for a, b in segments], lag, dim=d)
    eig = res.tica.eigenvalues.to_host()[:d]
    np.testing.assert_allclose(eig[:2], ref["eigenvalues"][:2], rtol=1e-9)
    np.testing.assert_allclose(np.sort(np.abs(eig)), np.sort(np.abs(ref["eigenvalues"][:d])), rtol=1e-7)
    Y = res.projected.to_host()
    Yo = npport.tica_transform(ref, Xp)
    for c in range(2):   # the resolved slow modes (the rest are a near-degenerate noise cluster)
        s = np.sign(np.dot(Y[:, c], Yo[:, c]))
        np.testing.assert_allclose(s * Y[:, c], Yo[:, c], atol=1e-7 * np.abs(Yo[:, c]).max())
    lab = res.labels.to_host()
    np.testing.assert_array_equal(lab, cport.kmeans_assign(Y, res.centers.to_host()))
    C, p = cport.count_transitions(lab, k, lag, segments=segments)
    np.testing.assert_array_equal(res.counts.to_host(), C)
    assert int(res.extras["pairs"].to_host()[0]) == p == sum(len(q) - lag for q in parts)
    est = pipe.estimate(res.counts, n_its=4, lag=float(lag))
    _, ts_ref = npport.its_from_counts(C.astype(float), lag, 4)
    ok = np.isfinite(ts_ref)
    np.testing.assert_allclose(est["spectrum"]["its_ts"][0][ok], ts_ref[ok], rtol=1e-6)


def test_c4_chignolin_ca_distances_k200_lagscan(engine, golden):
    """C4: chignolin, 45 C-alpha pair distances, k = 200, implied-timescale scan over lags."""
    g = golden("featurizer.npz")
    n = 40_000
    base = g["chig_xyz"][:18]                                   # the 18 NMR models, cyclically
    xyz = _tiled_frames(base, n, seed=1234) + _slow_walk(n, base.shape[1], seed=9)
    pairs = g["chig_pairs"]
    X = engine.featurize(engine.to_device(xyz), pairs=pairs)
    Xh = X.to_host()
    np.testing.assert_allclose(Xh, npport.distances(xyz.astype(np.float64), pairs), rtol=2e-6, atol=1e-7)
    pipe = MSMPipeline(engine)
    k = 200
    labels, centers, _ = pipe.cluster(X, k, seed=11, max_iter=15)
    lab = labels.to_host()
    np.testing.assert_array_equal(lab, cport.kmeans_assign(Xh.astype(np.float64), centers.to_host()))
    lags = [1, 2, 3, 5, 8, 12, 20, 35, 50]
    counts, npairs = engine.count_transitions_lagscan(labels, k, lags)
    Ch = counts.to_host()
    for i, lag in enumerate(lags):
        C, p = cport.count_transitions(lab, k, lag)
        np.testing.assert_array_equal(Ch[i], C)
        assert int(npairs.to_host()[i]) == p
    # batched ITS over the scan: one connected T per lag, one solve
    Ts, ns = [], []
    for i in range(len(lags)):
        tm = engine.transition_matrix(counts.view((k, k), np.int64, offset_elems=i * k * k), mode=1)
        Ts.append(tm["T"].to_host())
        ns.append(int(tm["n_active"].to_host()[0]))
    spec = engine.spectrum(engine.to_device(np.stack(Ts)), n=engine.to_device(np.asarray(ns, np.int32)), n_its=3,
                           lags=[float(v) for v in lags])
    for i, lag in enumerate(lags):
        _, ts_ref = npport.its_from_counts(Ch[i].astype(float), lag, 3)
        ok = np.isfinite(ts_ref)
        np.testing.assert_allclose(spec["its_ts"][i][ok], ts_ref[ok], rtol=1e-6)


def test_c5_wide_features_k2000(engine):
    """C5 shape at reduced N: 256 features, k = 2000, clustered in the raw feature space (the config
    names no TICA dimension) -- multi-tile centre staging, global-atomic member sums."""
    n, F, k, lag = 12_000, 256, 2000, 10
    X = _gen.correlated_series(n, F, seed=77)
    pipe = MSMPipeline(engine)
    xd = engine.to_device(X)
    labels, centers, state = pipe.cluster(xd, k, seed=5, max_iter=4)
    lab = labels.to_host()
    np.testing.assert_array_equal(lab, cport.kmeans_assign(X.astype(np.float64), centers.to_host()))
    counts, pairs = pipe.count(labels, k, lag)
    C, p = cport.count_transitions(lab, k, lag)
    np.testing.assert_array_equal(counts.to_host(), C)
    assert int(pairs.to_host()[0]) == p
    T = engine.transition_matrix(counts, mode=0)["T"].to_host()
    np.testing.assert_array_equal(T, npport.normalise_counts(C.astype(float)))


def test_c5_shape_at_size_labels_on_a_sample(engine):
    """C5 near its per-GPU size (300 K x 256, k = 2000; the full 1.25 M differs only in the number of frame groups a
    workgroup walks): every label of 4000 sampled frames against the pinned fp64 arg-min of the C oracle, the rest through a
    size-independent property -- the distance the device reports for a frame is the distance to the centre it names,
    and no centre of a sampled subset is nearer."""
    n, F, k = 300_000, 256, 2000
    rng = np.random.default_rng(9)
    X = _gen.correlated_series(n, F, seed=99)
    centers_h = X[np.sort(rng.choice(n, k, replace=False))].astype(np.float64) + 1e-3 * rng.normal(size=(k, F))
    xd, cd = engine.to_device(X), engine.to_device(centers_h)
    md = engine.empty((n,), np.float64)
    lab = engine.kmeans_assign(xd, cd, mindist=md).to_host()
    dist = md.to_host()
    idx = np.sort(rng.choice(n, 4000, replace=False))
    np.testing.assert_array_equal(lab[idx], cport.kmeans_assign(X[idx].astype(np.float64), centers_h))
    assert lab.min() >= 0 and lab.max() < k
    # reported distance = squared distance to the named centre (1e-9 relative), on another 20000 frames
    jdx = np.sort(rng.choice(n, 20_000, replace=False))
    d_named = ((X[jdx].astype(np.float64) - centers_h[lab[jdx]]) ** 2).sum(1)
    np.testing.assert_allclose(dist[jdx], d_named, rtol=1e-9, atol=1e-9)
    # and a random subset of 64 centres holds none that is nearer
    sub = rng.choice(k, 64, replace=False)
    Xj = X[jdx].astype(np.float64)
    d_sub = ((Xj * Xj).sum(1)[:, None] + (centers_h[sub] ** 2).sum(1)[None, :] - 2.0 * Xj @ centers_h[sub].T).min(1)
    assert np.all(d_sub >= d_named * (1 - 1e-9) - 1e-9)
