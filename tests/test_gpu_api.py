"""Operator-API shims (pmarlo_amd.features / .analysis / .markov_state_model) on the GPU.
The cases follow the reference's own tests (cited per test) so the two suites read alike."""

from __future__ import annotations

import json

import numpy as np
import pytest

from oracle import cport, npport
from tests import _gen
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _make_dataset(train, val=None, test=None):
    splits = {"train": {"X": np.asarray(train, dtype=np.float64)}}
    if val is not None:
        splits["val"] = {"X": np.asarray(val, dtype=np.float64)}
    if test is not None:
        splits["test"] = {"X": np.asarray(test, dtype=np.float64)}
    return {"splits": splits}


# ---- pmarlo.analysis (tests/analysis/test_discretize.py) -----------------------------
def test_prepare_msm_discretization_kmeans_assigns_all_splits(engine):
    from pmarlo_amd.analysis import prepare_msm_discretization

    train = np.array([[0.0, 0.0], [4.0, 4.0], [0.1, -0.1], [4.2, 3.9]])
    val = np.array([[0.05, 0.05], [4.1, 4.1]])
    test = np.array([[0.2, -0.05], [4.05, 4.02]])
    dataset = _make_dataset(train, val=val, test=test)
    dataset["splits"]["train"]["segments"] = [{"length": train.shape[0]}]
    result = prepare_msm_discretization(dataset, n_microstates=2, lag_time=1, random_state=0)
    assert set(result.assignments) == {"train", "val", "test"}
    assert result.counts.shape == (2, 2)
    for name, arr in result.assignments.items():
        assert arr.shape[0] == dataset["splits"][name]["X"].shape[0] and arr.dtype == np.int32
    assert result.feature_schema["n_features"] == 2
    assert result.segment_lengths["train"] == [4] and result.segment_strides["train"] == [1]
    assert result.counted_pairs["train"] == result.expected_pairs["train"] == 3
    assert result.fingerprint["expected_pairs"] == 3 and result.fingerprint["counted_pairs"] == 3
    art = dataset["__artifacts__"]
    for key in ("feature_stats", "state_assignments", "segment_lengths", "expected_pairs", "counted_pairs",
                "segment_strides"):
        assert key in art
    assert art["state_assignments"]["train"] == {"n_assigned": 4, "total": 4}
    for mask in result.assignment_masks.values():
        assert mask.dtype == np.bool_ and mask.all()
    # the two blobs are separated: frames 0,2 share a state, frames 1,3 the other
    a = result.assignments["train"]
    assert a[0] == a[2] != a[1] == a[3]
    assert result.assignments["val"][0] == a[0] and result.assignments["val"][1] == a[1]


def test_prepare_msm_expected_pairs_use_segment_stride_metadata(engine):
    from pmarlo_amd.analysis import expected_pairs, prepare_msm_discretization

    train = np.array([[0.0, 0.0], [0.1, -0.1], [0.2, 0.05], [0.3, -0.2]])
    dataset = _make_dataset(train)
    dataset["splits"]["train"]["segments"] = [{"length": 4, "stride": 2}]
    dataset["splits"]["train"]["feature_schema"] = {"names": ["feature_0", "feature_1"], "n_features": 2}
    result = prepare_msm_discretization(dataset, n_microstates=2, lag_time=1, random_state=0)
    assert result.segment_strides["train"] == [2]
    assert result.expected_pairs["train"] == expected_pairs([4], 1, [2]) == result.fingerprint["expected_pairs"]


def test_weighted_counts_use_starting_frame_weights(engine):
    from pmarlo_amd.analysis import prepare_msm_discretization

    train = np.array([[0.0, 0.0], [0.05, -0.05], [5.0, 5.0], [5.1, 5.2]])
    weights = np.array([1.0, 0.5, 2.0, 3.0])
    result = prepare_msm_discretization(_make_dataset(train), n_microstates=2, lag_time=1,
                                        frame_weights={"train": weights}, random_state=0)
    labels = result.assignments["train"]
    expected = np.zeros_like(result.counts)
    for idx in range(labels.size - 1):
        expected[labels[idx], labels[idx + 1]] += weights[idx]
    assert np.allclose(result.counts, expected)
    assert result.counted_pairs["train"] == result.expected_pairs["train"] == 3


def test_discretize_dataset_golden_parity_with_reference_centres(engine):
    """Full discretize_dataset result of the REFERENCE (3 splits, 2 segments, weights) reproduced
    when its fitted centres are supplied: labels bit-exact, counts / T to summation order."""
    from pmarlo_amd.analysis import prepare_msm_discretization

    g = np.load(GOLDEN / "discretize.npz")
    meta = json.loads((GOLDEN / "discretize.json").read_text())
    ds = {"splits": {
        "train": {"X": g["train"], "segments": [{"length": 250}, {"start": 250, "stop": 600, "stride": 2}]},
        "val": {"X": g["val"]}, "test": {"X": g["test"]}}}
    res = prepare_msm_discretization(ds, n_microstates=8, lag_time=2, random_state=3,
                                     frame_weights={"train": g["weights"]}, centers=g["centers"])
    for split in ("train", "val", "test"):
        np.testing.assert_array_equal(res.assignments[split], g[f"a_{split}"])
    np.testing.assert_allclose(res.fingerprint["scaler"]["mean"], meta["scaler"]["mean"], rtol=1e-13)
    np.testing.assert_allclose(res.fingerprint["scaler"]["std"], meta["scaler"]["std"], rtol=1e-12)
    np.testing.assert_allclose(res.counts, g["counts"], rtol=1e-12)
    np.testing.assert_allclose(res.transition_matrix, g["transition_matrix"], rtol=1e-12)
    np.testing.assert_allclose(res.diag_mass, float(g["diag_mass"]), rtol=1e-12)
    np.testing.assert_allclose(res.state_counts, g["state_counts"], rtol=1e-12)
    assert res.segment_lengths == meta["segment_lengths"] and res.segment_strides == meta["segment_strides"]
    assert res.counted_pairs == meta["counted_pairs"] and res.expected_pairs == meta["expected_pairs"]
    fp = {k: v for k, v in res.fingerprint.items() if k != "scaler"}
    assert fp == meta["fingerprint"]
    assert sorted(ds["__artifacts__"].keys()) == meta["artifacts_keys"]


def test_discretize_validation_errors(engine):
    from pmarlo_amd.analysis import ValidationError, discretize_dataset

    X = np.random.default_rng(0).normal(size=(50, 3))
    with pytest.raises(ValueError):
        discretize_dataset(_make_dataset(X), lag_time=0)
    bad = X.copy()
    bad[:, 1] = 1.0
    with pytest.raises(ValidationError) as ei:
        discretize_dataset(_make_dataset(bad), n_microstates=3)
    assert ei.value.code == "cv_zero_std"
    with pytest.raises(ValueError):
        discretize_dataset({"splits": {}})
    with pytest.raises(ValueError):
        discretize_dataset(_make_dataset(X), n_microstates=3, frame_weights={"train": np.ones(7)})


# ---- reduction (tests/unit/.../test_reduction.py:43-60) -------------------------------
def test_tica_reduce_matches_restated_deeptime(engine):
    from pmarlo_amd.markov_state_model import tica_reduce

    X = _gen.correlated_series(20_000, 12, seed=21)
    Y = tica_reduce(X, lag=10, n_components=3)
    Yo = npport.tica_reduce(X, lag=10, n_components=3)
    assert Y.shape == (20_000, 3) and Y.dtype == np.float64 and Y.flags.c_contiguous
    for c in range(3):  # up to column sign, atol 1e-6 as in the reference test
        s = np.sign(np.dot(Y[:, c], Yo[:, c]))
        np.testing.assert_allclose(s * Y[:, c], Yo[:, c], atol=1e-6)


def test_maybe_apply_tica_drops_lag_frames_per_trajectory(engine):
    """_features.py:181-231: dims clamped to [2,5], last `lag` frames of each trajectory dropped."""
    from pmarlo_amd.markov_state_model.reduction import tica_fit_transform_trajectories

    lens = [600, 600, 600]
    feats = np.vstack([_gen.correlated_series(n, 12, seed=i) for i, n in enumerate(lens)])
    Y, model = tica_fit_transform_trajectories(feats, lens, n_components_hint=9, lag=5)
    assert Y.shape == (sum(lens) - 3 * 5, 5)
    Xs = [feats[600 * i:600 * (i + 1)].astype(np.float64) for i in range(3)]
    ref = npport.tica_fit(Xs, 5, dim=5)
    np.testing.assert_allclose(model.eigenvalues.to_host()[:5], ref["eigenvalues"][:5], rtol=1e-8)


# ---- clustering (tests/unit/.../test_cluster_micro.py, perf determinism) ----------------
def test_cluster_microstates_contract(engine):
    from pmarlo_amd.markov_state_model import ClusteringResult, cluster_microstates

    Y, _ = _gen.gaussian_clusters(5, 800, 10, seed=42)
    Y = Y[np.random.default_rng(0).permutation(Y.shape[0])]
    res = cluster_microstates(Y, n_states=5, random_state=7)
    assert isinstance(res, ClusteringResult) and res.output_shape == (res.n_states,)
    assert res.labels.shape == (4000,) and res.labels.min() == 0 and res.labels.max() == res.n_states - 1
    for j in range(res.n_states):  # centres are member means (clustering.py:364-392)
        np.testing.assert_allclose(res.centers[j], Y[res.labels == j].mean(axis=0), rtol=1e-12, atol=1e-12)
    again = cluster_microstates(Y, n_states=5, random_state=7)
    np.testing.assert_array_equal(res.labels, again.labels)  # determinism under a fixed seed
    with pytest.raises(TypeError):
        cluster_microstates(Y, n_states=5, bogus=1)
    with pytest.raises(ValueError):
        cluster_microstates(Y[:, 0], n_states=5)
    empty = cluster_microstates(np.zeros((0, 3)), n_states=4)
    assert empty.n_states == 0 and empty.labels.size == 0


def test_cluster_microstates_takes_the_reference_keywords(engine):
    """The keyword set of the reference (clustering.py:236-262 _SUPPORTED_KWARGS) with its errors: every valid
    reference call must be a valid call here."""
    from pmarlo_amd.markov_state_model import cluster_microstates

    Y, true_centers = _gen.gaussian_clusters(6, 500, 4, seed=5)
    Y = Y[np.random.default_rng(1).permutation(Y.shape[0])]
    base = cluster_microstates(Y, n_states=6, random_state=3, max_iter=50, tolerance=1e-6)
    # n_jobs / progress: accepted and without effect on the device estimator; init_strategy / metric: the defaults
    same = cluster_microstates(Y, method="kmeans", n_states=6, random_state=3, max_iter=50, tolerance=1e-6, n_jobs=4,
                               progress=None, init_strategy="kmeans++", metric="euclidean")
    np.testing.assert_array_equal(base.labels, same.labels)
    # the mini-batch estimator is a different estimator (tests/test_gpu_kmeanspp.py); on six separated blobs it finds
    # the same partition
    mb = cluster_microstates(Y, method="minibatchkmeans", n_states=6, random_state=3, max_iter=5, tolerance=1e-6,
                             batch_size=256)
    assert mb.n_states == 6
    remap = {int(a): int(b) for a, b in zip(mb.labels, base.labels)}
    np.testing.assert_array_equal(np.array([remap[int(a)] for a in mb.labels]), base.labels)
    # initial_centers (the reference's spelling) starts the fit there: the true centres recover the blobs
    seeded = cluster_microstates(Y, n_states=6, initial_centers=true_centers, max_iter=20)
    assert seeded.n_states == 6
    for j in range(6):
        assert np.linalg.norm(seeded.centers[j] - true_centers[j]) < 0.1
    np.testing.assert_array_equal(seeded.labels, cluster_microstates(Y, n_states=6, init_centers=true_centers,
                                                                      max_iter=20).labels)
    # fixed_seed replaces random_state as the seed
    a = cluster_microstates(Y, n_states=6, random_state=None, fixed_seed=11)
    b = cluster_microstates(Y, n_states=6, random_state=11)
    np.testing.assert_array_equal(a.labels, b.labels)
    assert cluster_microstates(Y, n_states=6, fixed_seed=True).n_states == 6
    # restarts (n_init) keep the lowest inertia: never worse than the first seed alone
    r = cluster_microstates(Y, n_states=6, random_state=3, n_init=4)
    def inertia(res):
        return float(((Y - res.centers[res.labels]) ** 2).sum())
    assert inertia(r) <= inertia(cluster_microstates(Y, n_states=6, random_state=3)) * (1 + 1e-12)
    # the reference's errors
    with pytest.raises(TypeError, match="batch_size"):
        cluster_microstates(Y, method="kmeans", n_states=6, batch_size=64)
    with pytest.raises(ValueError, match="batch_size"):
        cluster_microstates(Y, method="auto", n_states=6, batch_size=64)        # auto picks kmeans at this size
    with pytest.raises(ValueError, match="n_init cannot be combined"):
        cluster_microstates(Y, n_states=6, n_init=3, fixed_seed=5)
    with pytest.raises(TypeError, match="fixed_seed"):
        cluster_microstates(Y, n_states=6, fixed_seed="7")
    with pytest.raises(ValueError, match="positive"):
        cluster_microstates(Y, n_states=0)
    with pytest.raises(TypeError, match="Unsupported clustering parameters"):
        cluster_microstates(Y, n_states=6, n_clusters=6)
    # fewer occupied states than requested: labels are densified, centres follow
    dup = np.repeat(np.array([[0.0, 0.0], [5.0, 5.0], [9.0, 0.0]]), 50, axis=0)
    few = cluster_microstates(dup, n_states=8, random_state=0)
    assert few.n_states == 3 and sorted(np.unique(few.labels)) == [0, 1, 2] and few.centers.shape == (3, 2)


# ---- estimation (test_markov_state_model.py:18-61, test_deeptime_backend.py:95-109) -------
def test_build_msm_invariants_and_oracle(engine):
    from pmarlo_amd.markov_state_model import build_msm, count_transitions

    rng = np.random.default_rng(42)
    T_true = np.array([[0.7, 0.2, 0.1], [0.2, 0.7, 0.1], [0.1, 0.2, 0.7]])
    traj = np.empty(60_000, dtype=int)
    traj[0] = 0
    u = rng.random(traj.size)
    cdf = np.cumsum(T_true, axis=1)
    for i in range(1, traj.size):
        traj[i] = int(np.searchsorted(cdf[traj[i - 1]], u[i]))
    dtrajs = [traj[:25_000], traj[25_000:]]
    C = count_transitions(dtrajs, 3, lag=1)
    want = sum(cport.count_transitions(d, 3, 1)[0] for d in dtrajs)
    np.testing.assert_array_equal(C, want.astype(float))
    est = build_msm(dtrajs, 3, lag_time=1)
    ref = npport.ml_msm(C)
    np.testing.assert_allclose(est.transition_matrix, ref["transition_matrix"], rtol=1e-13)
    np.testing.assert_allclose(est.stationary_distribution, ref["stationary_distribution"], rtol=1e-9)
    np.testing.assert_allclose(est.transition_matrix.sum(axis=1), 1.0, atol=1e-12)
    np.testing.assert_allclose(est.transition_matrix.T @ est.stationary_distribution, est.stationary_distribution,
                               atol=1e-12)
    np.testing.assert_allclose(est.transition_matrix, T_true, atol=2e-2)
    est2 = build_msm(dtrajs, 3, lag_time=2)  # Chapman-Kolmogorov, atol 2e-2 as in the reference
    np.testing.assert_allclose(est2.transition_matrix, est.transition_matrix @ est.transition_matrix, atol=2e-2)
    assert est.free_energies.min() == 0.0
    # invalid labels split trajectories (no pair bridges them)
    bad = traj[:1000].copy()
    bad[500] = -1
    Cb = count_transitions([bad], 3, lag=3)
    wb = cport.count_transitions(bad[:500], 3, 3)[0] + cport.count_transitions(bad[501:], 3, 3)[0]
    np.testing.assert_array_equal(Cb, wb.astype(float))


# ---- ITS (test_two_state_msm.py:6-22, test_its_plateau.py:6-40, test_its_math.py) -----------
def test_implied_timescales_two_state_and_plateau(engine, golden):
    from pmarlo_amd.markov_state_model import compute_implied_timescales, safe_timescales

    rng = np.random.default_rng(0)
    n = 200_000
    flips = rng.random(n) < 0.1
    traj = (np.cumsum(flips) % 2).astype(int)   # T = [[0.9, 0.1], [0.1, 0.9]] -> lambda2 = 0.8
    res = compute_implied_timescales([traj], 2, lag_times=[1, 2, 3, 5, 8], n_timescales=1, n_samples=0)
    np.testing.assert_array_equal(res.lag_times, [1, 2, 3, 5, 8])
    assert res.timescales.shape == (5, 1) and res.timescales_ci.shape == (5, 1, 2)
    t_true = -1.0 / np.log(0.8)
    assert abs(res.timescales[0, 0] - t_true) / t_true < 0.10          # reference tolerance: 10 %
    assert np.all(np.abs(res.timescales[:, 0] - t_true) / t_true < 0.20)  # plateau over lags: 20 %
    np.testing.assert_allclose(res.rates, 1.0 / res.timescales)
    for i, lag in enumerate(res.lag_times):  # deterministic definition == oracle, 1e-5 relative (north_star)
        C, _ = cport.count_transitions(traj.astype(np.int32), 2, int(lag))
        _, ts_ref = npport.its_from_counts(C, int(lag), 1)
        np.testing.assert_allclose(res.timescales[i], ts_ref, rtol=1e-8)
    g = golden("timescales.npz")
    np.testing.assert_array_equal(safe_timescales(25, g["eig"]), g["ts_lag25"])
    np.testing.assert_array_equal(safe_timescales(5.0, g["ceig"]), g["cts_lag5"])
    assert compute_implied_timescales([traj[:3]], 2, lag_times=[5]).lag_times.size == 0


# ---- features (registry + featurize_trajectory) ------------------------------------------------
_PDB = """\
ATOM      1  N   ALA A   1      -0.677  -1.230  -0.491  1.00  0.00           N
ATOM      2  CA  ALA A   1      -0.001   0.064  -0.491  1.00  0.00           C
ATOM      3  C   ALA A   1       1.499  -0.110  -0.491  1.00  0.00           C
ATOM      4  O   ALA A   1       2.030  -1.227  -0.502  1.00  0.00           O
ATOM      5  N   GLY A   2       2.250   1.000  -0.400  1.00  0.00           N
ATOM      6  CA  GLY A   2       3.700   0.950  -0.300  1.00  0.00           C
ATOM      7  C   GLY A   2       4.300   2.300   0.100  1.00  0.00           C
ATOM      8  O   GLY A   2       3.600   3.300   0.200  1.00  0.00           O
ATOM      9  N   SER A   3       5.600   2.350   0.350  1.00  0.00           N
ATOM     10  CA  SER A   3       6.300   3.600   0.700  1.00  0.00           C
ATOM     11  C   SER A   3       7.800   3.400   0.900  1.00  0.00           C
ATOM     12  O   SER A   3       8.300   2.300   0.800  1.00  0.00           O
END
"""


def test_feature_registry_and_featurize_trajectory(engine, tmp_path):
    from pmarlo_amd.features import featurize_trajectory, get_feature, parse_feature_spec, register_feature
    from pmarlo_amd.io import Trajectory, load_pdb

    p = tmp_path / "tri.pdb"
    p.write_text(_PDB)
    base = load_pdb(p)
    assert base.n_atoms == 12 and base.topology.n_residues == 3
    np.testing.assert_array_equal(base.topology.phi_indices(), [[2, 4, 5, 6], [6, 8, 9, 10]])
    np.testing.assert_array_equal(base.topology.psi_indices(), [[0, 1, 2, 4], [4, 5, 6, 8]])
    rng = np.random.default_rng(1234)
    xyz = np.tile(base.xyz, (500, 1, 1)) + rng.normal(0, 0.02, size=(500, 12, 3)).astype(np.float32)
    traj = Trajectory(xyz, base.topology)
    X = featurize_trajectory(traj, "phi_psi")
    quads = np.vstack([base.topology.phi_indices(), base.topology.psi_indices()])
    np.testing.assert_allclose(X, npport.dihedrals(traj.xyz, quads), atol=3e-5)
    D = featurize_trajectory(traj, "ca_distances")
    np.testing.assert_allclose(D, npport.distances(traj.xyz, [[1, 5], [1, 9], [5, 9]]), rtol=3e-6)
    with pytest.raises(ValueError):
        featurize_trajectory(traj, "nope")
    phi_psi = get_feature("phi_psi")
    Xr = phi_psi.compute(traj)
    np.testing.assert_array_equal(Xr, X)
    assert phi_psi.is_periodic().tolist() == [True] * 4 and phi_psi.labels[0].startswith("phi:res")
    name, kw = parse_feature_spec("distance([1, 5])")
    d = get_feature(name).compute(traj, **kw)
    np.testing.assert_allclose(d[:, 0], D[:, 0].astype(float), rtol=0, atol=0)
    assert get_feature("distance").is_periodic().tolist() == [False]
    with pytest.raises(ValueError):
        get_feature("distance").compute(traj, indices=[1, 99])
    with pytest.raises(KeyError):
        get_feature("unknown_feature")

    class Mine:  # last registration wins (base.py:30-33)
        name = "phi_psi"

        def compute(self, traj, **kw):
            return np.zeros((traj.n_frames, 1))

        def is_periodic(self):
            return np.zeros(1, bool)

    register_feature(Mine())
    assert get_feature("PHI_PSI").compute(traj).shape == (500, 1)
    register_feature(phi_psi)


def test_compute_msm_features_layouts_and_tica_step(engine, golden):
    """FeaturesMixin.compute_features (S/markov_state_model/_features.py:23-97, 131-171, 181-231):
    block layout [cos phi | sin phi | cos psi | sin psi], every-third-C-alpha distance pairs, stride,
    TICA clamp to [2, 5] with the last lag frames of every trajectory dropped."""
    from pmarlo_amd.io import Topology, Trajectory
    from pmarlo_amd.markov_state_model import ca_distance_pairs, compute_msm_features

    g = golden("featurizer.npz")
    rng = np.random.default_rng(5)
    # alanine dipeptide: 22 atoms, phi = [4, 6, 8, 14], psi = [6, 8, 14, 16]
    names = ["X"] * 22
    resid = np.zeros(22, dtype=int)
    for idx, (nm, r) in {4: ("C", 0), 6: ("N", 1), 8: ("CA", 1), 14: ("C", 1), 16: ("N", 2)}.items():
        names[idx], resid[idx] = nm, r
    resid[15:] = 2
    resid[5:15] = 1
    top = Topology(names, ["ALA"] * 22, resid, ["A"] * 22)
    assert top.phi_indices().tolist() == [[4, 6, 8, 14]] and top.psi_indices().tolist() == [[6, 8, 14, 16]]
    trajs = []
    for n in (400, 250):
        xyz = np.tile(g["ala_xyz"][:1], (n, 1, 1)).astype(np.float32)
        trajs.append(Trajectory(xyz + rng.normal(0, 0.03, size=xyz.shape).astype(np.float32), top))
    out = compute_msm_features(trajs, "phi_psi", feature_stride=2)
    assert out.raw_frames == 650 and out.strided_frames == 325 and out.traj_lengths == [200, 125]
    x0 = trajs[0].xyz[::2].astype(np.float64)
    phi = npport.dihedrals(x0, [[4, 6, 8, 14]])[:, 0]
    psi = npport.dihedrals(x0, [[6, 8, 14, 16]])[:, 0]
    want = np.column_stack([np.cos(phi), np.sin(phi), np.cos(psi), np.sin(psi)])
    np.testing.assert_allclose(out.features[:200], want, atol=3e-5)
    # TICA step: hint 9 -> 5 dims is impossible with 4 features -> rank-limited to <= 4; lag frames dropped per trajectory
    red = compute_msm_features(trajs, "phi_psi", tica_lag=3, tica_components=2)
    assert red.features.shape == (400 - 3 + 250 - 3, 2) and red.traj_lengths == [397, 247]
    red0 = compute_msm_features(trajs, "phi_psi", tica_lag=0, tica_components=2)
    assert red0.features.shape == (650, 2)                      # lag 0: fitted at lag 1, nothing dropped
    # every third C-alpha, j >= i + 3, capped
    ca = list(range(0, 60, 2))                                    # 30 "C-alpha" atoms
    pairs = ca_distance_pairs(ca, None)
    assert pairs[0].tolist() == [0, 6] and pairs[1].tolist() == [0, 12] and len(pairs) == 45
    assert len(ca_distance_pairs(ca, 7)) == 7
    with pytest.raises(ValueError):
        ca_distance_pairs([3], None)
    with pytest.raises(ValueError):
        compute_msm_features(trajs, "bogus")


def test_pca_reduce_vs_reference_golden_and_oracle(engine, golden):
    """pca_reduce / reduce_features(method="pca") (S/markov_state_model/reduction.py:43-74, 152-197)."""
    from pmarlo_amd.markov_state_model import pca_reduce, reduce_features

    g = golden("pca.npz")
    for name, kw in (("scaled", dict(n_components=3, scale=True)), ("raw", dict(n_components=5, scale=False))):
        want = g[f"{name}_Y"]
        got = pca_reduce(g["X"], **kw)
        assert got.shape == want.shape and got.dtype == np.float64
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-8 * np.abs(want).max())
    X = _gen.correlated_series(200_000, 64, seed=3)
    got = reduce_features(X, method="pca", n_components=4)
    want = npport.pca_reduce(X.astype(np.float64), 4)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-8 * np.abs(want).max())
    with pytest.raises(ValueError):
        reduce_features(X[:100], method="umap")
    with pytest.raises(ValueError):
        pca_reduce(X[:100], n_components=65)


# ---- VAMP reduction (reduction.vamp_reduce; deeptime absent -> oracle restatement) ---------------
@pytest.mark.parametrize("n,F,lag,dim,scale", [(20_000, 8, 5, 3, True), (6_000, 20, 1, 4, False), (3_000, 3, 10, 5, True)])
def test_vamp_reduce_vs_oracle(engine, n, F, lag, dim, scale):
    from pmarlo_amd.markov_state_model import reduce_features, vamp_reduce

    X = _gen.correlated_series(n, F, seed=n + F).astype(np.float64)
    X *= np.linspace(0.5, 3.0, F)[None, :]
    X[::97, 1] = np.nan                                         # imputed to the column mean by _preprocess
    got = vamp_reduce(X, lag=lag, n_components=dim, scale=scale)
    want, s = npport.vamp_reduce(X, lag=lag, n_components=dim, scale=scale)
    assert got.shape == want.shape == (n, min(dim, F))
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-8 * np.abs(want).max())
    # the instantaneous window is whitened by construction
    C = np.cov(got[:-lag].T, bias=True)
    np.testing.assert_allclose(C, np.eye(got.shape[1]), atol=1e-8)
    # leading singular values of an AR(1) mixture: the latent autocorrelations at this lag
    if (n, F) == (20_000, 8):
        assert abs(s[0] - 0.985 ** lag) < 0.06 and abs(s[1] - 0.95 ** lag) < 0.08 and s[2] < 0.2
    np.testing.assert_array_equal(reduce_features(X, method="vamp", n_components=dim, lag=lag, scale=scale), got)
    with pytest.raises(ValueError):
        vamp_reduce(X[:5], lag=10)


# ---- validate_features (analysis/validation.py:89-172; golden made with the reference) ------------
def test_validate_features_vs_reference_golden(engine, golden):
    from pmarlo_amd.analysis.validation import ValidationError, validate_features

    X = golden("validate.npz")["X"]
    want = json.loads((GOLDEN / "validate.json").read_text())
    bad = X.copy(); bad[3, 1] = np.nan; bad[7, 4] = np.inf
    none = X[:6].copy(); none[:, 0] = np.nan
    flat = X.copy(); flat[:, 2] = 4.25
    cases = {"ok": (X, ["a", "b", "c"]), "non_finite": (bad, None),
             "no_finite_rows": (none, ["p", "q", "r", "s", "t", "u"]), "zero_std": (flat, None)}
    for name, (M, names) in cases.items():
        try:
            stats, code = validate_features(M, names), None
        except ValidationError as exc:
            stats, code = exc.stats, exc.code
        ref = want[name]
        assert code == ref["code"], name
        assert sorted(stats) == sorted(ref["stats"]), name
        for key, val in ref["stats"].items():
            if key in ("means", "stds", "mins", "maxs"):
                np.testing.assert_allclose(np.asarray(stats[key], float), np.asarray(val, float), rtol=1e-12,
                                           atol=1e-12, equal_nan=True, err_msg=f"{name}:{key}")
            else:
                assert stats[key] == val, (name, key)
    with pytest.raises(ValueError):
        validate_features(np.zeros(5), None)


def test_device_view_keeps_its_allocation_alive(engine):
    """Regression test for a use-after-free (commit d301652): a view of a device array outlived the base array it
    pointed into (the stationary vector inside compute_committor), so the allocator could hand the memory to the next
    call.  A view now holds a reference to its base: dropping the base must not free the memory under the view."""
    import gc

    base = engine.to_device(np.arange(1024, dtype=np.float64))
    view = base.view((16,), offset_elems=512)
    chained = view.view((4,), offset_elems=8)
    del base
    gc.collect()
    # churn the allocator: same-sized allocations would reuse a freed block at once
    junk = [engine.to_device(np.full(1024, -1.0)) for _ in range(32)]
    np.testing.assert_array_equal(view.to_host(), np.arange(512, 528, dtype=np.float64))
    np.testing.assert_array_equal(chained.to_host(), np.arange(520, 524, dtype=np.float64))
    del junk


def test_pca_reduce_with_batches_equals_incremental_pca(engine):
    """pca_reduce(batch_size=...) = _preprocess + sklearn IncrementalPCA(n_components, batch_size).fit_transform
    (S/markov_state_model/reduction.py:69-73), batch by batch on the device: against sklearn itself on the same
    preprocessed data (the restated _preprocess is pinned by the golden fixtures)."""
    from sklearn.decomposition import IncrementalPCA

    from pmarlo_amd.markov_state_model import pca_reduce

    rng = np.random.default_rng(4)
    X = _gen.correlated_series(12_345, 16, seed=8).astype(np.float64) * rng.uniform(0.5, 20.0, size=16) + rng.normal(size=16)
    X[17, 3] = np.nan                                   # imputed by the column mean in _preprocess
    for batch, k, scale in ((1000, 4, True), (4096, 3, False), (12_345, 5, True), (777, 16, True)):
        got = pca_reduce(X, n_components=k, batch_size=batch, scale=scale)
        Xp = npport.preprocess(X, scale=scale)
        want = IncrementalPCA(n_components=k, batch_size=batch).fit_transform(Xp.copy())
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-7 * np.abs(want).max())
    with pytest.raises(ValueError):
        pca_reduce(X[:100], n_components=12, batch_size=10)
