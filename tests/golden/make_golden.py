"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Run in the build container only (the reference tree does not travel):

    PYTHONPATH=/root/reference/src python tests/golden/make_golden.py

Re-running reproduces every committed array exactly, except the scikit-learn k-means centres
(discretize.npz `centers`, kmeans.npz `ar_centers`), which come back within one ulp (threaded sums).

Every array written here is either an input (seeded synthetic data, or
coordinates parsed from the reference's data/*.pdb) or the output of a reference
function on that input.  No reference source text is stored.  The functions
exercised are the importable half of the path:
pmarlo.analysis.{discretize,counting,debug_export,msm}, reduction._preprocess,
trainer_api._estimate_top_eigenvalues, utils.safe_timescales, validation.ck_rule and the
torch feature extractor.
"""

from __future__ import annotations

import hashlib
import json
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(REF / "src"))

from pmarlo.analysis import discretize as ref_disc  # noqa: E402
from pmarlo.analysis.counting import expected_pairs as ref_expected_pairs  # noqa: E402
from pmarlo.analysis.debug_export import compute_analysis_debug  # noqa: E402
from pmarlo.analysis.msm import prepare_msm_discretization  # noqa: E402
from pmarlo.features.deeptica.core.trainer_api import _estimate_top_eigenvalues  # noqa: E402
from pmarlo.features.deeptica.ts_feature_extractor import (  # noqa: E402
    build_feature_extractor_module,
    canonicalize_feature_spec,
)
from pmarlo.markov_state_model.reduction import _preprocess  # noqa: E402
from pmarlo.markov_state_model.utils import safe_timescales  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---- generators (same recipes as the reference's perf/integration tests) ----
def correlated_series(n_frames, n_features, seed):
    """tests/perf/test_tica_perf.py:65-81 recipe (AR(1) latents), loop form."""
    rng = np.random.default_rng(seed)
    latent = np.zeros((n_frames, 3))
    for t in range(1, n_frames):
        latent[t, 0] = 0.985 * latent[t - 1, 0] + rng.normal(scale=0.05)
        latent[t, 1] = 0.950 * latent[t - 1, 1] + rng.normal(scale=0.08)
        latent[t, 2] = rng.normal(scale=0.5)
    mixing = rng.normal(scale=0.7, size=(3, n_features))
    noise = rng.normal(scale=0.05, size=(n_frames, n_features))
    return (latent @ mixing + noise).astype(np.float32)


def gaussian_clusters(n_clusters, per, d, seed):
    """tests/perf/test_discretize_assignment_perf.py:30-48 recipe."""
    rng = np.random.default_rng(seed)
    centers = rng.normal(loc=0.0, scale=5.0, size=(n_clusters, d))
    data = np.vstack([c + rng.normal(scale=0.2, size=(per, d)) for c in centers])
    return data.astype(np.float64), centers


def two_well_labels(n, tau_corr, seed):
    """tests/integration/test_msm_synthetic.py:11-26 recipe."""
    rng = np.random.default_rng(seed)
    x = np.zeros(n)
    for t in range(1, n):
        f = -4.0 * x[t - 1] * (x[t - 1] ** 2 - 1.0)
        x[t] = x[t - 1] + f / tau_corr + np.sqrt(2.0 / tau_corr) * rng.normal()
    return (x > 0).astype(np.int32)


def read_pdb_models(path):
    models, cur, names = [], [], []
    first = True
    for line in open(path):
        rec = line[:6]
        if rec in ("ATOM  ", "HETATM"):
            cur.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
            if first:
                names.append(line[12:16].strip())
        elif rec == "ENDMDL":
            models.append(cur)
            cur = []
            first = False
    if cur:
        models.append(cur)
    return np.asarray(models, dtype=np.float64) / 10.0, names  # nm


# ---- 1. counts ---------------------------------------------------------------
def golden_counts():
    rng = np.random.default_rng(7)
    out = {}
    lab = rng.integers(0, 7, size=5000).astype(np.int32)
    lab[rng.random(5000) < 0.03] = -1
    w = rng.random(5000)
    segs = [(0, 1200), (1200, 1210), (1210, 3700), (3700, 5000)]
    cases = [
        ("plain", dict(lag_time=3), None),
        ("segs", dict(lag_time=5, segments=segs), None),
        ("stride", dict(lag_time=4, segments=segs, stride=3), None),
        ("weighted", dict(lag_time=2, segments=segs), w),
        ("lag_ge_len", dict(lag_time=1300, segments=segs), None),
    ]
    out["labels"] = lab
    out["weights"] = w
    out["segments"] = np.asarray(segs, dtype=np.int64)
    for name, kw, ww in cases:
        c, p = ref_disc._weighted_counts(lab, n_states=7, weights=ww, **kw)
        out[f"{name}_counts"] = c
        out[f"{name}_pairs"] = np.int64(p)
    out["state_counts"] = ref_disc._compute_state_counts(lab, n_states=7)
    out["normalised_plain"] = ref_disc._normalise_counts(out["plain_counts"])
    # expected_pairs table
    ep_in = [([10, 5, 0, 7], 3, 1), ([100], 10, [4]), ([5, 5], 5, 1), ([12, 9, 30], 2, [1, 2, 5])]
    out["expected_pairs"] = np.asarray([ref_expected_pairs(a, b, c) for a, b, c in ep_in], dtype=np.int64)
    # double well of tests/integration/test_msm_synthetic.py:59-68
    z = two_well_labels(20000, 800, 11)
    dbg = compute_analysis_debug({"dtrajs": [z.astype(int)]}, lag=400, count_mode="sliding")
    out["two_well_labels"] = z.astype(np.uint8)
    out["two_well_counts"] = np.asarray(dbg.counts)
    out["two_well_total_pairs"] = np.int64(dbg.summary["total_pairs"])
    np.savez_compressed(OUT / "counts.npz", **out)


# ---- 2. preprocess + 3. eigenvalues ------------------------------------------
def golden_preprocess_tica():
    rng = np.random.default_rng(5)
    X = (rng.normal(size=(1000, 8)) * rng.uniform(0.5, 3.0, size=8) + rng.normal(size=8)).astype(np.float32)
    X[rng.random(X.shape) < 0.01] = np.nan
    X[:, 6] = 2.5  # constant column: std 0 -> scale 1
    out = {"pre_X": X, "pre_out_scale": _preprocess(X, scale=True), "pre_out_noscale": _preprocess(X, scale=False)}
    Y = correlated_series(4000, 8, 21)
    lag = 10
    idx_t = np.arange(0, Y.shape[0] - lag)
    idx_tau = idx_t + lag
    Yp = _preprocess(Y, scale=True)
    ev = _estimate_top_eigenvalues(Yp, idx_t, idx_tau, SimpleNamespace(n_out=8))
    out.update(tica_X=Y, tica_lag=np.int64(lag), tica_top_eigs=np.asarray(ev))
    # the reference's own benchmark input of this estimator (tests/perf/test_tica_perf.py:216-235): raw AR(1)
    # series, N = 20 000, F = 8, seed 21, lag 10.  Only its hash is stored: tests regenerate it (tests/_gen.py).
    P = correlated_series(20_000, 8, 21)
    ip = np.arange(0, P.shape[0] - lag)
    evp = _estimate_top_eigenvalues(P, ip, ip + lag, SimpleNamespace(n_out=8))
    # arbitrary (non-run) pairs on the same input: every third frame against frame + 7 and + 13 alternately
    it = np.arange(0, P.shape[0] - 20, 3)
    itau = it + np.where(np.arange(it.size) % 2 == 0, 7, 13)
    evq = _estimate_top_eigenvalues(P, it, itau, SimpleNamespace(n_out=8))
    # float32 outputs make the reference work in float32 throughout (centring, both products, both LAPACK calls):
    # its own answer then carries ~1e-6 of single-precision error.  The same series as float64 gives the values an
    # fp64 implementation can be held to at 1e-9.
    P64 = P.astype(np.float64)
    evp64 = _estimate_top_eigenvalues(P64, ip, ip + lag, SimpleNamespace(n_out=8))
    evq64 = _estimate_top_eigenvalues(P64, it, itau, SimpleNamespace(n_out=8))
    out.update(perf_input_sha=np.frombuffer(bytes.fromhex(sha(P)), np.uint8), perf_lag=np.int64(lag),
               perf_top_eigs=np.asarray(evp), perf_pairs_eigs=np.asarray(evq),
               perf_top_eigs_f64=np.asarray(evp64), perf_pairs_eigs_f64=np.asarray(evq64))
    np.savez_compressed(OUT / "tica.npz", **out)


# ---- 4. k-means discretizer ---------------------------------------------------
def golden_kmeans():
    out = {}
    X, _ = gaussian_clusters(6, 200, 4, 21)
    disc = ref_disc._KMeansDiscretizer(6, random_state=0)
    disc.fit(X)
    out.update(km_X=X, km_mean=disc.scaler_mean_, km_std=disc.scaler_std_, km_centers=disc.centers,
               km_labels=disc.transform(X))
    Xq, _ = gaussian_clusters(6, 50, 4, 52)
    out.update(km_Xq=Xq, km_labels_q=disc.transform(Xq))
    # AR(1) data, more centres than blobs (near-tie rich)
    Y = correlated_series(3000, 4, 3).astype(np.float64)
    d2 = ref_disc._KMeansDiscretizer(40, random_state=1)
    d2.fit(Y)
    out.update(ar_X=Y, ar_mean=d2.scaler_mean_, ar_std=d2.scaler_std_, ar_centers=d2.centers,
               ar_labels=d2.transform(Y))
    # MiniBatch branch (n*f >= 5e6): input regenerated from the seed in the test
    Xb, _ = gaussian_clusters(50, 10000, 10, 99)
    d3 = ref_disc._KMeansDiscretizer(50, random_state=0)
    d3.fit(Xb)
    lb = d3.transform(Xb)
    out.update(mb_input_sha=np.frombuffer(bytes.fromhex(sha(Xb)), dtype=np.uint8),
               mb_mean=d3.scaler_mean_, mb_std=d3.scaler_std_, mb_centers=d3.centers,
               mb_labels_head=lb[:8192], mb_labels_sha=np.frombuffer(bytes.fromhex(sha(lb)), dtype=np.uint8),
               mb_bincount=np.bincount(lb, minlength=50).astype(np.int64),
               mb_is_minibatch=np.bool_(type(d3.model).__name__ == "MiniBatchKMeans"))
    np.savez_compressed(OUT / "kmeans.npz", **out)


# ---- 6. discretize_dataset ----------------------------------------------------
def golden_discretize():
    rng = np.random.default_rng(17)
    train, _ = gaussian_clusters(5, 120, 3, 31)
    perm = rng.permutation(train.shape[0])
    # make it a time series with metastability: sort blocks then jitter order locally
    train = train[np.argsort(perm // 40, kind="stable")]
    val, _ = gaussian_clusters(5, 20, 3, 31)
    test, _ = gaussian_clusters(5, 10, 3, 31)
    w = rng.uniform(0.5, 1.5, size=train.shape[0])
    ds = {"splits": {
        "train": {"X": train, "segments": [{"length": 250}, {"start": 250, "stop": 600, "stride": 2}]},
        "val": {"X": val},
        "test": {"X": test},
    }}
    res = prepare_msm_discretization(ds, n_microstates=8, lag_time=2, random_state=3,
                                     frame_weights={"train": w})
    out = dict(train=train, val=val, test=test, weights=w, centers=res.centers, counts=res.counts,
               transition_matrix=res.transition_matrix, diag_mass=np.float64(res.diag_mass),
               state_counts=res.state_counts, counts_before_prune=res.counts_before_prune,
               a_train=res.assignments["train"], a_val=res.assignments["val"], a_test=res.assignments["test"])
    meta = dict(segment_lengths=res.segment_lengths, segment_strides=res.segment_strides,
                counted_pairs=res.counted_pairs, expected_pairs=res.expected_pairs,
                fingerprint={k: v for k, v in res.fingerprint.items() if k != "scaler"},
                scaler=res.fingerprint["scaler"], lag_time=res.lag_time, cluster_mode=res.cluster_mode,
                artifacts_keys=sorted(ds["__artifacts__"].keys()))
    np.savez_compressed(OUT / "discretize.npz", **out)
    (OUT / "discretize.json").write_text(json.dumps(meta, indent=1, sort_keys=True))


# ---- 7. safe_timescales --------------------------------------------------------
def golden_timescales():
    eig = np.array([0.2, 0.5, 0.8, 0.95, 1.0, 1.2, 0.0, -0.3, 1e-14, 1 - 1e-14, np.nan])
    ceig = np.array([0.95 * np.exp(1j * np.pi / 4), 0.5 + 0.5j, -0.2 + 0j, 0.9 + 1e-12j, 1.1j])
    np.savez_compressed(OUT / "timescales.npz", eig=eig, ts_lag25=safe_timescales(25, eig),
                        ceig=ceig, cts_lag5=safe_timescales(5.0, ceig),
                        eig2d=np.array([[0.2, 0.5], [0.8, 0.95]]),
                        ts2d=safe_timescales(25, np.array([[0.2, 0.5], [0.8, 0.95]])))


# ---- 1'. featurizer -------------------------------------------------------------
def golden_featurizer():
    import torch

    out = {}
    chig, names = read_pdb_models(REF / "data" / "chignolin.pdb")
    ca = [i for i, nm in enumerate(names) if nm == "CA"]
    pairs = [(ca[i], ca[j]) for i in range(len(ca)) for j in range(i + 1, len(ca))]
    rng = np.random.default_rng(1234)
    xyz = np.tile(chig, (2, 1, 1)).astype(np.float32)
    xyz[18:] += rng.normal(0, 0.02, size=xyz[18:].shape).astype(np.float32)
    spec = canonicalize_feature_spec({"use_pbc": False, "features": [
        {"type": "distance", "atoms": list(p)} for p in pairs]})
    ext = build_feature_extractor_module(spec).eval()
    box = torch.eye(3)
    d = np.stack([ext(torch.from_numpy(f), box).numpy() for f in xyz])
    out.update(chig_xyz=xyz, chig_ca=np.asarray(ca, np.int32), chig_pairs=np.asarray(pairs, np.int32), chig_dist=d)

    ala, _ = read_pdb_models(REF / "data" / "alanine-dipeptide.pdb")
    quads = [[4, 6, 8, 14], [6, 8, 14, 16]]
    trip = [[4, 6, 8], [6, 8, 14], [8, 14, 16]]
    axyz = np.tile(ala, (64, 1, 1)).astype(np.float32)
    axyz[1:] += rng.normal(0, 0.03, size=axyz[1:].shape).astype(np.float32)
    spec2 = canonicalize_feature_spec({"use_pbc": False, "features": (
        [{"type": "dihedral", "atoms": q} for q in quads] + [{"type": "angle", "atoms": t} for t in trip])})
    ext2 = build_feature_extractor_module(spec2).eval()
    f2 = np.stack([ext2(torch.from_numpy(f), box).numpy() for f in axyz])
    out.update(ala_xyz=axyz, ala_quads=np.asarray(quads, np.int32), ala_triplets=np.asarray(trip, np.int32),
               ala_dihedrals=f2[:, :2], ala_angles=f2[:, 2:])
    # unit-cube known answer of tests/features/deeptica/test_ts_feature_extractor.py:44-74
    cube = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [1, 1, 1]], dtype=np.float32)
    spec3 = canonicalize_feature_spec({"use_pbc": False, "features": [
        {"type": "distance", "atoms": [0, 1]}, {"type": "angle", "atoms": [0, 1, 2]},
        {"type": "dihedral", "atoms": [0, 1, 2, 3]}]})
    out.update(cube_xyz=cube[None], cube_feats=build_feature_extractor_module(spec3).eval()(
        torch.from_numpy(cube), box).numpy())
    np.savez_compressed(OUT / "featurizer.npz", **out)


def golden_ck():
    """validation/ck_rule.py: ck_error, _multinomial_rms_se, decide_ck (both modes) on seeded
    row-stochastic matrices: a metastable chain whose lag-k matrices are its exact powers plus
    sampling noise, so some factors pass and some fail."""
    from pmarlo.validation import ck_rule

    rng = np.random.default_rng(77)
    n = 12
    base = rng.random((n, n)) * 0.02
    for b in range(3):
        base[4 * b:4 * b + 4, 4 * b:4 * b + 4] += rng.random((4, 4)) + 0.5
    P = base / base.sum(axis=1, keepdims=True)
    out = {"P": P}
    P_taus, P_ktaus, rows = {}, {}, {}
    for k, noise in ((2, 0.0), (3, 0.002), (4, 0.2), (5, 0.01)):
        Pk = np.linalg.matrix_power(P, k) + noise * rng.random((n, n))
        Pk /= Pk.sum(axis=1, keepdims=True)
        counts = rng.integers(40, 4000, size=n).astype(float)
        if k == 3:
            counts[2] = 0.0          # exercises the N_i <= 0 -> 1 rule
        P_taus[k], P_ktaus[k], rows[k] = P, Pk, counts
        out[f"Pk_{k}"] = Pk
        out[f"rows_{k}"] = counts
        out[f"err_{k}"] = ck_rule.ck_error(P, Pk, k)
        out[f"se_{k}"] = ck_rule._multinomial_rms_se(Pk, counts)
    for mode in ("ess_adjusted", "absolute"):
        cfg = ck_rule.CKConfig(mode=mode, k_steps=(2, 3, 4, 5))
        dec = ck_rule.decide_ck(P_taus, P_ktaus, rows, cfg)
        out[f"{mode}_pass_fraction"] = dec.pass_fraction
        out[f"{mode}_passed"] = dec.passed
        out[f"{mode}_per_lag"] = np.array([[k, v["error"], v["threshold"], v["noise_rms"], v["pass"]]
                                           for k, v in sorted(dec.per_lag.items())])
    np.savez_compressed(OUT / "ck.npz", **out)


def golden_pca():
    """reduction.pca_reduce (sklearn PCA behind _preprocess), scale on / off, with missing values."""
    from pmarlo.markov_state_model.reduction import pca_reduce as ref_pca

    X = correlated_series(3000, 12, seed=31).astype(np.float64) * np.linspace(0.5, 3.0, 12)[None, :]
    X[np.random.default_rng(5).random(X.shape) < 0.004] = np.nan
    out = {"X": X}
    for name, kw in (("scaled", dict(n_components=3, scale=True)), ("raw", dict(n_components=5, scale=False))):
        out[f"{name}_Y"] = ref_pca(X.copy(), **kw)
    np.savez_compressed(OUT / "pca.npz", **out)


def golden_fes():
    """analysis/fes.py: compute_weighted_fes, grid and KDE, weighted and not (whitening off: the
    dataset carries no DeepTICA metadata)."""
    from pmarlo.analysis import fes as ref_fes

    rng = np.random.default_rng(404)
    n = 12000
    nb = n // 3
    blobs = np.concatenate([rng.normal([-1.0, 0.5, 0.0], [0.35, 0.6, 0.05], size=(nb, 3)),
                            rng.normal([1.2, -0.4, 0.0], [0.5, 0.3, 0.05], size=(nb, 3))])
    flat = rng.uniform([-2.0, -1.5, -0.1], [2.5, 1.8, 0.1], size=(n - 2 * nb, 3))   # keeps every grid bin populated
    X = np.clip(np.concatenate([blobs, flat]), [-2.0, -1.5, -1.0], [2.5, 1.8, 1.0])
    X = X[rng.permutation(n)]
    X = X[:, [2, 0, 1]]                      # the low-variance column first: exercises the component selection
    w = rng.gamma(2.0, 1.0, size=n)
    out = {"X": X, "w": w}
    cases = {
        "grid_u": dict(method="grid", bins=12, min_count_per_bin=1, weights=None),
        "grid_w": dict(method="grid", bins=(10, 14), min_count_per_bin=70, weights=w),
        "grid_s": dict(method="grid", bins=16, min_count_per_bin=25, weights=None),
        "kde_u": dict(method="kde", bins=16, bandwidth="scott", weights=None),
        "kde_w": dict(method="kde", bins=(12, 20), bandwidth="silverman", weights=w),
        "kde_f": dict(method="kde", bins=9, bandwidth=0.3, weights=w, temperature_K=350.0),
    }
    for name, kw in cases.items():
        res = ref_fes.compute_weighted_fes({"splits": {"train": {"X": X.copy()}}}, apply_whitening=False, **kw)
        out[f"{name}_hist"] = res["histogram"]
        out[f"{name}_xedges"] = res["xedges"]
        out[f"{name}_yedges"] = res["yedges"]
        out[f"{name}_F"] = res["free_energy"]
        md = res["metadata"]
        out[f"{name}_sel"] = np.asarray(md["selected_components"])
        if "bandwidth" in md:
            b = md["bandwidth"]
            out[f"{name}_bw"] = np.array([b["x"], b["y"], b["effective_sample_size"], b["total_weight"]])
        else:
            out[f"{name}_smoothed"] = np.array([md["smoothed_bins"]])
    np.savez_compressed(OUT / "fes.npz", **out)


def golden_grid():
    """_GridDiscretizer (analysis/discretize.py:517-593): first-appearance state numbering on a
    regular grid; the test split reaches cells the training split never visits, a constant column
    exercises the lo == hi branch, and values sit exactly on edges."""
    rng = np.random.default_rng(23)
    out = {}
    train = rng.normal(size=(3000, 3)) * np.array([1.0, 2.0, 0.5])
    train[:, 1] = np.round(train[:, 1], 1)
    test = rng.normal(size=(800, 3)) * np.array([2.5, 4.0, 1.5])
    test[::97, 0] = np.nan
    test[5::131, 2] = np.inf
    g = ref_disc._GridDiscretizer(target_states=60)
    g.fit(train)
    test[3::59, 0] = g.edges[0][2]          # exactly on an interior edge
    test[7::61, 1] = g.edges[1][-1]         # exactly the training maximum
    out.update(a_train=train, a_test=test, a_edges=np.stack(g.edges), a_lab_train=g.transform(train),
               a_lab_test=g.transform(test), a_lab_train_again=g.transform(train), a_centers=g.centers,
               a_n_states=np.int64(len(g.mapping)))
    flat = rng.normal(size=(500, 2))
    flat[:, 1] = 1.25
    g2 = ref_disc._GridDiscretizer(target_states=25)
    g2.fit(flat)
    out.update(b_train=flat, b_edges=np.stack(g2.edges), b_lab=g2.transform(flat))
    # whole entry point, grid mode
    tr, _ = gaussian_clusters(5, 120, 3, 31)
    tr = tr[np.argsort(rng.permutation(tr.shape[0]) // 40, kind="stable")]
    va, _ = gaussian_clusters(5, 20, 3, 32)
    ds = {"splits": {"train": {"X": tr}, "val": {"X": va * 1.5}}}
    res = prepare_msm_discretization(ds, cluster_mode="grid", n_microstates=27, lag_time=2)
    out.update(c_train=tr, c_val=va * 1.5, c_a_train=res.assignments["train"], c_a_val=res.assignments["val"],
               c_counts=res.counts, c_T=res.transition_matrix, c_state_counts=res.state_counts,
               c_centers=res.centers)
    np.savez_compressed(OUT / "grid.npz", **out)


def golden_its_helpers():
    """ITSMixin._detect_timescale_plateau, _its_alpha_tail_bounds and _its_default_lag_times
    (markov_state_model/_its.py; the module imports without deeptime, these methods never reach it)."""
    from pmarlo.markov_state_model._its import ITSMixin

    host = ITSMixin.__new__(ITSMixin)
    rng = np.random.default_rng(3)
    out = {"default_lags": np.asarray(host._its_default_lag_times(None)),
           "tails": np.asarray([host._its_alpha_tail_bounds(c) for c in (0.95, 0.9, 0.5)])}
    lags = np.asarray([1, 2, 3, 5, 8, 10, 15, 20, 30, 40, 50, 75], dtype=float)
    series = []
    for case in range(12):
        ts = 40.0 * (1.0 - np.exp(-lags / rng.uniform(1.0, 12.0))) + rng.normal(scale=rng.uniform(0.0, 1.5), size=lags.size)
        if case % 4 == 1:
            ts[rng.integers(0, lags.size)] = np.nan
        if case % 4 == 2:
            ts[:3] = -1.0
        if case == 11:
            ts[:] = np.nan
        series.append(np.stack([ts, 0.3 * ts], axis=1))
    series = np.asarray(series)
    wins = []
    for ts in series:
        for m, eps in ((2, 0.05), (3, 0.1), (4, 0.2), (1, 0.0)):
            w = host._detect_timescale_plateau(lags, ts, m, eps)
            wins.append((np.nan, np.nan) if w is None else w)
    out.update(plateau_lags=lags, plateau_series=series, plateau_windows=np.asarray(wins, dtype=float))
    np.savez_compressed(OUT / "its_helpers.npz", **out)


def golden_validate():
    """analysis/validation.validate_features: statistics of a clean matrix, and the error codes / statistics of
    the three failure modes (non-finite entries, no finite row, zero-variance column)."""
    from pmarlo.analysis.validation import ValidationError, validate_features

    rng = np.random.default_rng(9)
    X = rng.normal(size=(400, 5)) * np.array([1.0, 3.0, 0.2, 10.0, 1e-3]) + np.array([0.0, 5.0, -2.0, 100.0, 1.0])
    cases = {"ok": (X, ["a", "b", "c"])}
    bad = X.copy(); bad[3, 1] = np.nan; bad[7, 4] = np.inf
    cases["non_finite"] = (bad, None)
    none = X[:6].copy(); none[:, 0] = np.nan
    cases["no_finite_rows"] = (none, ["p", "q", "r", "s", "t", "u"])
    flat = X.copy(); flat[:, 2] = 4.25
    cases["zero_std"] = (flat, None)
    out = {}
    for name, (M, names) in cases.items():
        try:
            out[name] = {"stats": validate_features(M, names), "code": None}
        except ValidationError as exc:
            out[name] = {"stats": exc.stats, "code": exc.code}
    np.savez_compressed(OUT / "validate.npz", X=X)
    (OUT / "validate.json").write_text(json.dumps(out, indent=1, sort_keys=True, default=float))


def golden_free_energy():
    """markov_state_model/free_energy.py (imports without deeptime): periodic_kde_2d, free_energy_from_density,
    generate_1d_pmf on seeded samples."""
    from pmarlo.markov_state_model import free_energy as fe

    rng = np.random.default_rng(31)
    n = 2500
    tx = np.concatenate([rng.vonmises(-2.8, 4.0, n // 2), rng.vonmises(1.0, 8.0, n - n // 2)])
    ty = np.concatenate([rng.vonmises(3.0, 6.0, n // 2), rng.vonmises(-0.5, 3.0, n - n // 2)])
    out = dict(tx=tx, ty=ty, kde_default=fe.periodic_kde_2d(tx, ty),
               kde_fine=fe.periodic_kde_2d(tx, ty, bw=(0.2, 0.5), gridsize=(30, 70)))
    dens = out["kde_default"].copy()
    dens[3, :5] = 0.0
    mask = dens < 0.002
    out.update(dens=dens, mask=mask, F_plain=fe.free_energy_from_density(dens, 300.0),
               F_mask=fe.free_energy_from_density(dens, 310.0, mask=mask),
               F_inpaint=fe.free_energy_from_density(dens, 310.0, mask=mask, inpaint=True),
               F_tiny=fe.free_energy_from_density(dens, 300.0, tiny=0.01))
    cv = np.concatenate([rng.normal(-1.0, 0.4, 4000), rng.normal(1.5, 0.7, 3000)])
    for name, kw in (("pmf_plain", dict(bins=60)), ("pmf_smooth", dict(bins=80, smoothing_sigma=1.5)),
                     ("pmf_periodic", dict(bins=36, periodic=True, range_=(-np.pi, np.pi), smoothing_sigma=0.8)),
                     ("pmf_range", dict(bins=25, range_=(-0.5, 2.0), temperature=350.0))):
        data = tx if name == "pmf_periodic" else cv
        r = fe.generate_1d_pmf(data, **kw)
        out.update({f"{name}_F": r.F, f"{name}_edges": r.edges, f"{name}_counts": r.counts})
    out["cv"] = cv
    np.savez_compressed(OUT / "free_energy.npz", **out)


def golden_debug():
    """analysis/debug_export.compute_analysis_debug: full summaries (counts, visits, SCC, dwell times, warnings)
    for sliding and strided counting, unassigned frames, an isolated and a sink-only state."""
    rng = np.random.default_rng(77)

    def chain(n, k, stay, seed):
        r = np.random.default_rng(seed)
        x = np.zeros(n, dtype=int)
        for t in range(1, n):
            x[t] = x[t - 1] if r.random() < stay else r.integers(k)
        return x

    a = chain(9000, 8, 0.9, 1)
    b = chain(4000, 8, 0.7, 2)
    b[[1200, 2900]] = -1            # two unassigned frames (more would trip the reference's pair-count check)
    c = chain(700, 6, 0.5, 3)
    c[-1] = 9                      # state 9 is only ever entered (zero row), state 8 never seen
    cases = {"sliding": ([a, b, c], 3, "sliding"), "strided": ([a, b], 4, "strided"),
             "tiny": ([c[:200], np.array([], dtype=int), np.full(5, -1)], 2, "sliding")}
    arrays, summaries = {}, {}
    for name, (dtrajs, lag, mode) in cases.items():
        dbg = compute_analysis_debug({"dtrajs": dtrajs}, lag=lag, count_mode=mode)
        summaries[name] = dbg.to_summary_dict()
        arrays[f"{name}_counts"] = dbg.counts
        for i, d in enumerate(dtrajs):
            arrays[f"{name}_dtraj{i}"] = np.asarray(d, dtype=np.int64)
        summaries[name]["_lag"], summaries[name]["_mode"], summaries[name]["_n_dtrajs"] = lag, mode, len(dtrajs)
    np.savez_compressed(OUT / "debug.npz", **arrays)
    (OUT / "debug.json").write_text(json.dumps(summaries, indent=1, sort_keys=True, default=float))


def golden_fes2d():
    """markov_state_model/free_energy.generate_2d_fes: adaptive / fixed grids, a periodic torus, given ranges,
    and the three smoothing modes."""
    from pmarlo.markov_state_model.free_energy import generate_2d_fes

    rng = np.random.default_rng(41)
    n = 6000
    a = np.concatenate([rng.normal(-1.0, 0.4, n // 2), rng.normal(1.2, 0.6, n - n // 2)])
    b = np.concatenate([rng.normal(0.5, 0.3, n // 2), rng.normal(-0.8, 0.5, n - n // 2)])
    a[:3] = [9.0, -8.0, 7.5]                                # outliers: cut by the 1 % / 99 % crop
    phi = rng.vonmises(-1.2, 3.0, n)
    psi = rng.vonmises(2.4, 1.5, n)
    cases = {
        "adaptive": (a, b, dict(bins=(40, 40))),
        "fixed": (a, b, dict(bins=(30, 50), grid_strategy="fixed", min_count=2)),
        "ranges": (a, b, dict(bins=(25, 25), ranges=((-2.0, 2.5), (-2.0, 1.5)), grid_strategy="fixed", temperature=330.0)),
        "torus": (phi, psi, dict(bins=(36, 36), periodic=(True, True), grid_strategy="fixed")),
        "half_torus": (phi, b, dict(bins=(20, 20), periodic=(True, False))),
        "auto": (a, b, dict(bins=(40, 40), fes_smoothing_mode="auto")),
        "always": (a, b, dict(bins=(40, 40), grid_strategy="fixed", config={"fes_smoothing_mode": "always", "fes_h0": 0.9})),
    }
    out = dict(a=a, b=b, phi=phi, psi=psi)
    for name, (u, v, kw) in cases.items():
        r = generate_2d_fes(u, v, **kw)
        md = r.metadata
        out.update({f"{name}_F": r.F, f"{name}_xedges": r.xedges, f"{name}_yedges": r.yedges,
                    f"{name}_density": md["counts"], f"{name}_mask": md["mask"],
                    f"{name}_shape": np.asarray(md["grid_shape"]), f"{name}_empty": np.float64(md["empty_bins_fraction"]),
                    f"{name}_applied": np.float64(md["smoothing"]["applied_fraction"])})
    np.savez_compressed(OUT / "fes2d.npz", **out)


def golden_fes_calculator():
    """free_energy.FESCalculator.calculate_fes: MSM-reweighted FES in kT (weights = pi[microstate])."""
    from pmarlo.markov_state_model.free_energy import FESCalculator

    rng = np.random.default_rng(53)
    k = 12
    pi = rng.dirichlet(np.ones(k) * 2.0)
    proj = [rng.normal(size=(3000, 3)) * [1.0, 0.5, 2.0], rng.normal(size=(2000, 3)) + [1.5, -0.5, 0.0]]
    dtr = [rng.integers(0, k, 3000), rng.integers(0, k + 2, 2000)]        # two labels beyond pi: filtered
    msm = SimpleNamespace(stationary_distribution=pi)
    calc = FESCalculator({"temperature": 310.0})
    out = dict(pi=pi, proj0=proj[0], proj1=proj[1], d0=dtr[0], d1=dtr[1])
    for name, kw in (("default", dict(bins=40)), ("dims", dict(bins=25, dim_x=2, dim_y=0, max_energy_cap_kt=None)),
                     ("cap", dict(bins=30, max_energy_cap_kt=3.0))):
        grid, F = calc.calculate_fes(proj, msm, dtrajs=dtr, **kw)
        out.update({f"{name}_xx": grid[0], f"{name}_yy": grid[1], f"{name}_F": F})
    np.savez_compressed(OUT / "fes_calculator.npz", **out)


def golden_msm_fes():
    """markov_state_model/_fes.FESMixin numerics called as unbound methods on a stand-in object (the module
    imports without deeptime; _map_stationary_to_frame_weights then takes its pi[state] fall-back)."""
    from pmarlo.markov_state_model._fes import FESMixin

    rng = np.random.default_rng(61)
    k = 9
    pi = rng.dirichlet(np.ones(k))
    dtr = [rng.integers(0, k, 5000), rng.integers(0, k, 2500)]
    obj = SimpleNamespace(dtrajs=dtr, stationary_distribution=pi, lag_time=3)
    w = FESMixin._map_stationary_to_frame_weights(obj)
    phi = rng.vonmises(-1.0, 2.0, w.size) * 180.0 / np.pi
    psi = rng.vonmises(2.0, 1.0, w.size) * 180.0 / np.pi
    u = rng.normal(size=w.size) * 2.0
    v = rng.normal(size=w.size) + 0.3 * u
    out = dict(pi=pi, d0=dtr[0], d1=dtr[1], weights=w, phi=phi, psi=psi, u=u, v=v,
               bins=np.asarray([FESMixin._choose_bins(obj, t, b) for t, b in ((0, 30), (7500, 50), (7500, 44), (10 ** 6, 10), (90000, 58))]))
    for name, (a, b, ranges, per) in {"torsion": (phi, psi, [(-180.0, 180.0), (-180.0, 180.0)], True), "plain": (u, v, None, False)}.items():
        nb = FESMixin._choose_bins(obj, w.size, 50)
        H, xe, ye = FESMixin._compute_weighted_histogram(obj, a, b, w, nb, ranges, smooth_sigma=0.6, periodic=per)
        out.update({f"{name}_H": H, f"{name}_xe": xe, f"{name}_ye": ye, f"{name}_F": FESMixin._histogram_to_free_energy(obj, H, 300.0)})
    np.savez_compressed(OUT / "msm_fes.npz", **out)


def golden_fes_smoothing():
    """markov_state_model/fes_smoothing.py on a seeded count grid."""
    from pmarlo.markov_state_model import fes_smoothing as fs

    rng = np.random.default_rng(71)
    counts = rng.poisson(rng.gamma(0.6, 8.0, size=(24, 31))).astype(float)
    F = rng.normal(size=counts.shape) * 3.0 + 5.0
    mask, sd = fs.mark_bins_for_smoothing(counts, target_sd_kT=0.5, alpha=1e-6, kT=2.5)
    h = fs.adaptive_bandwidth(counts, h0=1.2, ess_ref=50.0, h_min=0.4, h_max=3.0)
    out = dict(counts=counts, F=F, mask=mask, sd=sd, h=h, smooth_all=fs.smooth_F_with_adaptive_gaussian(F, h),
               smooth_masked=fs.smooth_F_with_adaptive_gaussian(F, h, apply_mask=mask),
               smooth_grid=fs.smooth_F_with_adaptive_gaussian(F, h, sigma_grid=(0.3, 0.9, 2.5)),
               sd_default=fs.fes_uncertainty_sd_kT(counts))
    np.savez_compressed(OUT / "fes_smoothing.npz", **out)


def golden_whitening():
    """analysis/project_cv.apply_whitening_from_metadata (ml/deeptica/whitening.apply_output_transform) and
    analysis/fes.ensure_fes_inputs_whitened + compute_weighted_fes on a dataset with DeepTICA artifacts."""
    from pmarlo.analysis import fes as ref_fes
    from pmarlo.analysis.project_cv import apply_whitening_from_metadata

    rng = np.random.default_rng(83)
    n, d = 4000, 3
    A = rng.normal(size=(d, d))
    Y = rng.normal(size=(n, d)) @ A + np.array([2.0, -1.0, 0.5])
    mean = Y.mean(0) + rng.normal(scale=0.05, size=d)
    W = np.linalg.inv(np.linalg.cholesky(np.cov(Y.T))).T + rng.normal(scale=0.02, size=(d, d))
    out = dict(Y=Y, mean=mean, W=W)
    md = {"output_mean": mean.tolist(), "output_transform": W.tolist(), "output_transform_applied": "false"}
    out["whitened"], applied = apply_whitening_from_metadata(Y, md)
    out["applied"] = np.bool_(applied)
    out["again"], again = apply_whitening_from_metadata(Y, md)            # flag now set: returned unchanged
    out["again_applied"] = np.bool_(again)
    out["few"], _ = apply_whitening_from_metadata(Y[:2], {"output_mean": mean, "output_transform": W})   # n <= d: no batch whitening
    ds = {"X": Y.copy(), "splits": {"train": {"X": Y[:3000].copy()}, "val": {"X": Y[3000:].copy()}},
          "__artifacts__": {"mlcv_deeptica": {"output_mean": mean.tolist(), "output_transform": W.tolist()}}}
    res = ref_fes.compute_weighted_fes(ds, split="train", bins=14, method="kde")
    out.update(fes_F=res["free_energy"], fes_hist=res["histogram"], fes_xedges=res["xedges"], fes_yedges=res["yedges"],
               ds_X=np.asarray(ds["X"]), ds_train=np.asarray(ds["splits"]["train"]["X"]), ds_val=np.asarray(ds["splits"]["val"]["X"]))
    np.savez_compressed(OUT / "whitening.npz", **out)


# ---- 20. the reference's own trajectory fixtures (tests/_assets): a truncated copy + struct-level facts --------
def golden_real_assets():
    """tests/_assets/traj.dcd (100 frames x 3 350 atoms, unit cells, written by OpenMM) and 3gd8-fixed.pdb are the
    files the reference's own tests load through mdtraj (tests/conftest.py:208-229).  Stored: the header and the first
    four frames of the DCD byte for byte (frame count patched to 4), the PDB text, and what an independent parse with
    `struct` / column slicing says about them (header fields, unit cells, coordinates, atom / residue tables)."""
    import struct

    assets = REF / "tests" / "_assets"
    raw = (assets / "traj.dcd").read_bytes()
    nf_keep = 4
    blk = struct.unpack_from("<i", raw, 0)[0]
    assert blk == 84 and raw[4:8] == b"CORD"
    icntrl = struct.unpack_from("<9i", raw, 8)
    delta = struct.unpack_from("<f", raw, 8 + 36)[0]
    tail = struct.unpack_from("<10i", raw, 8 + 40)
    off = 4 + 84 + 4
    tsize = struct.unpack_from("<i", raw, off)[0]
    ntitle = struct.unpack_from("<i", raw, off + 4)[0]
    title = raw[off + 8:off + 8 + 80].split(b"\x00")[0].decode()
    off += 4 + tsize + 4
    assert struct.unpack_from("<i", raw, off)[0] == 4
    natoms = struct.unpack_from("<i", raw, off + 4)[0]
    off += 12
    has_cell = tail[0] == 1
    frame_bytes = (56 if has_cell else 0) + 3 * (4 * natoms + 8)
    assert len(raw) == off + icntrl[0] * frame_bytes
    cells, xyz = [], []
    for fr in range(nf_keep):
        o = off + fr * frame_bytes
        if has_cell:
            assert struct.unpack_from("<i", raw, o)[0] == 48
            cells.append(struct.unpack_from("<6d", raw, o + 4))
            o += 56
        cols = []
        for ax in range(3):
            assert struct.unpack_from("<i", raw, o)[0] == 4 * natoms
            cols.append(np.frombuffer(raw, dtype="<f4", count=natoms, offset=o + 4).copy())
            o += 4 * natoms + 8
        xyz.append(np.stack(cols, axis=1))
    xyz = np.asarray(xyz)                       # Angstrom, as stored
    keep = bytearray(raw[:off + nf_keep * frame_bytes])
    struct.pack_into("<i", keep, 8, nf_keep)    # NSET of the truncated copy
    pdb_text = (assets / "3gd8-fixed.pdb").read_bytes()
    names, resn, resseq, elems, chains = [], [], [], [], []
    for line in pdb_text.decode().splitlines():
        if line.startswith(("ATOM  ", "HETATM")):
            names.append(line[12:16].strip())
            resn.append(line[17:20].strip())
            chains.append(line[21])
            resseq.append(int(line[22:26]))
            elems.append(line[76:78].strip())
    res_change = np.r_[True, [(chains[i], resseq[i]) != (chains[i - 1], resseq[i - 1]) for i in range(1, len(names))]]
    res_index = np.cumsum(res_change) - 1
    np.savez_compressed(
        OUT / "real_assets.npz",
        dcd_bytes=np.frombuffer(bytes(keep), np.uint8), pdb_text=np.frombuffer(pdb_text, np.uint8),
        dcd_n_frames_total=np.int64(icntrl[0]), dcd_istart=np.int64(icntrl[1]), dcd_nsavc=np.int64(icntrl[2]),
        dcd_delta=np.float32(delta), dcd_has_cell=np.int64(has_cell), dcd_charmm_version=np.int64(tail[9]),
        dcd_natoms=np.int64(natoms), dcd_ntitle=np.int64(ntitle), dcd_title=np.frombuffer(title.encode(), np.uint8),
        dcd_cells_raw=np.asarray(cells), dcd_xyz_angstrom_head=xyz[:, :8], dcd_xyz_angstrom_tail=xyz[:, -8:],
        dcd_xyz_sums=xyz.astype(np.float64).sum(axis=1), dcd_xyz_sq_sums=(xyz.astype(np.float64) ** 2).sum(axis=1),
        pdb_natoms=np.int64(len(names)), pdb_nres=np.int64(res_index[-1] + 1),
        pdb_ca=np.flatnonzero(np.asarray(names) == "CA"), pdb_n_hydrogen=np.int64(sum(e == "H" for e in elems)),
        pdb_first_names=np.frombuffer(" ".join(names[:12]).encode(), np.uint8),
        pdb_first_resnames=np.frombuffer(" ".join(np.asarray(resn)[np.flatnonzero(res_change)[:8]]).encode(), np.uint8),
        pdb_resseq_first_last=np.asarray([resseq[0], resseq[-1]]), pdb_res_index_sha=np.frombuffer(
            bytes.fromhex(sha(res_index.astype(np.int64))), np.uint8))


if __name__ == "__main__":
    golden_real_assets()
    golden_whitening()
    golden_fes_smoothing()
    golden_msm_fes()
    golden_fes_calculator()
    golden_fes2d()
    golden_debug()
    golden_free_energy()
    golden_validate()
    golden_its_helpers()
    golden_grid()
    golden_pca()
    golden_fes()
    golden_ck()
    golden_counts()
    golden_preprocess_tica()
    golden_kmeans()
    golden_discretize()
    golden_timescales()
    golden_featurizer()
    for p in sorted(OUT.glob("*.np*")) + sorted(OUT.glob("*.json")):
        print(f"{p.name:24s} {p.stat().st_size / 1024:8.1f} KB")
