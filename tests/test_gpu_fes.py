"""Free-energy surfaces on the GPU (msm_weighted_stats, msm_hist2d, msm_smooth_sparse_bins,
msm_fes_finalize, msm_kde2d) against the golden vectors made by importing the reference's
analysis/fes.py, and against the oracle on other sizes.

Tolerances: unweighted histograms bit-exact (integer counts); weighted histograms 1e-12 relative
(2^e fixed point vs numpy's sequential sum); KDE densities 1e-11 relative (sum order, exp);
free energies 1e-9 absolute kJ/mol."""
import numpy as np
import pytest

from oracle import npport
from pmarlo_amd.analysis.fes import compute_weighted_fes, select_highest_variance_components

pytestmark = pytest.mark.gpu

CASES = [
    ("grid_u", dict(method="grid", bins=12, min_count_per_bin=1)),
    ("grid_w", dict(method="grid", bins=(10, 14), min_count_per_bin=70, weighted=True)),
    ("grid_s", dict(method="grid", bins=16, min_count_per_bin=25)),
    ("kde_u", dict(method="kde", bins=16, bandwidth="scott")),
    ("kde_w", dict(method="kde", bins=(12, 20), bandwidth="silverman", weighted=True)),
    ("kde_f", dict(method="kde", bins=9, bandwidth=0.3, weighted=True, temperature_K=350.0)),
]


@pytest.mark.parametrize("case,kw", CASES)
def test_compute_weighted_fes_vs_reference_golden(golden, case, kw):
    g = golden("fes.npz")
    kw = dict(kw)
    w = g["w"] if kw.pop("weighted", False) else None
    out = compute_weighted_fes({"splits": {"train": {"X": g["X"]}}}, weights=w, apply_whitening=False, **kw)
    md = out["metadata"]
    assert md["selected_components"] == g[f"{case}_sel"].tolist()
    assert md["weighted"] == (w is not None) and md["split"] == "train"
    np.testing.assert_allclose(out["xedges"], g[f"{case}_xedges"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(out["yedges"], g[f"{case}_yedges"], rtol=1e-13, atol=1e-13)
    if case == "grid_u":
        np.testing.assert_array_equal(out["histogram"], g[f"{case}_hist"])     # integer counts
    else:
        np.testing.assert_allclose(out["histogram"], g[f"{case}_hist"], rtol=1e-11)
    np.testing.assert_allclose(out["free_energy"], g[f"{case}_F"], rtol=0, atol=1e-9)
    if kw["method"] == "kde":
        b = md["bandwidth"]
        np.testing.assert_allclose([b["x"], b["y"], b["effective_sample_size"], b["total_weight"]], g[f"{case}_bw"],
                                   rtol=1e-12)
    else:
        assert md["smoothed_bins"] == int(g[f"{case}_smoothed"][0])


@pytest.mark.parametrize("n,bins,method", [(300_000, 64, "kde"), (1_000_000, (48, 100), "grid"), (50_001, 70, "kde")])
def test_fes_vs_oracle_large(n, bins, method):
    rng = np.random.default_rng(n)
    X = np.concatenate([np.clip(rng.normal([0, 0, 0, 0], [1.0, 0.01, 2.0, 0.5], size=(n - n // 5, 4)), -5.9, 5.9),
                        rng.uniform(-6, 6, size=(n // 5, 4)) * [1.0, 0.001, 1.0, 0.1]])   # uniform floor: no empty bins
    w = rng.gamma(1.5, 1.0, size=n)
    nk = 20_000 if method == "kde" else n     # the oracle's KDE materialises (bins x N) arrays
    ds = {"splits": {"a": {"X": X[:nk]}}, "frame_weights": {"a": w[:nk]}}
    got = compute_weighted_fes(ds, bins=bins, method=method, min_count_per_bin=2, apply_whitening=False)
    want = npport.weighted_fes(X[:nk], weights=w[:nk], bins=bins, method=method, min_count_per_bin=2)
    assert got["metadata"]["selected_components"] == want["metadata"]["selected_components"] == [2, 0]
    np.testing.assert_allclose(got["histogram"], want["histogram"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(got["free_energy"], want["free_energy"], rtol=0, atol=1e-8)
    if method == "kde" and n > nk:   # full-size run: finite, normalised density (integrates to ~1 on the padded grid)
        full = compute_weighted_fes({"splits": {"a": {"X": X}}}, weights=w, bins=bins, method="kde", apply_whitening=False)
        dx = np.diff(full["xedges"])[0] * np.diff(full["yedges"])[0]
        assert abs(full["histogram"].sum() * dx - 1.0) < 2e-2 and np.all(np.isfinite(full["free_energy"]))


def test_histogram_edge_semantics(engine):
    """last edge inclusive, values outside dropped, NaN dropped, searchsorted('right') on the edges"""
    xe = np.linspace(0.0, 1.0, 11)
    ye = np.array([0.0, 0.5, 1.0])
    pts = np.array([[0.0, 0.0], [1.0, 1.0], [0.3, 0.5], [0.30000000000000004, 0.25], [0.1 * 3, 0.75], [1.0000001, 0.5],
                    [-1e-9, 0.5], [np.nan, 0.5], [0.5, np.nan], [0.7, 0.4999999999999999]])
    h = engine.hist2d(engine.to_device(pts), (0, 1), xe, ye).to_host()
    ok = np.isfinite(pts).all(axis=1)
    want, _, _ = np.histogram2d(pts[ok, 0], pts[ok, 1], bins=[xe, ye])
    np.testing.assert_array_equal(h, want)
    assert h.sum() == 6


def test_errors_and_component_selection():
    rng = np.random.default_rng(0)
    X = rng.normal(size=(500, 3)) * [0.1, 5.0, 1.0]
    coords, sel = select_highest_variance_components(X)
    assert sel == [1, 2] and coords.shape == (500, 2)
    with pytest.raises(ValueError):
        select_highest_variance_components(np.ones((10, 3)))
    ds = {"splits": {"s": {"X": X}}}
    with pytest.raises(ValueError):
        compute_weighted_fes(ds, method="spline", apply_whitening=False)
    with pytest.raises(ValueError):
        compute_weighted_fes(ds, weights=-np.ones(500), apply_whitening=False)
    with pytest.raises(ValueError):
        compute_weighted_fes(ds, weights=np.ones(3), apply_whitening=False)
    with pytest.raises(ValueError):    # empty bins stay empty without smoothing -> the reference raises too
        compute_weighted_fes(ds, method="grid", bins=40, min_count_per_bin=0, apply_whitening=False)
    with pytest.raises(KeyError):
        compute_weighted_fes(ds, split="missing", apply_whitening=False)


def test_free_energy_module_vs_reference_golden(golden):
    """pmarlo.markov_state_model.free_energy (periodic_kde_2d, free_energy_from_density, generate_1d_pmf):
    golden vectors made by importing the reference; KDE 1e-11 (sum order, exp), histograms exact counts."""
    from pmarlo_amd.markov_state_model.free_energy import free_energy_from_density, generate_1d_pmf, periodic_kde_2d

    g = golden("free_energy.npz")
    np.testing.assert_allclose(periodic_kde_2d(g["tx"], g["ty"]), g["kde_default"], rtol=1e-11)
    np.testing.assert_allclose(periodic_kde_2d(g["tx"], g["ty"], bw=(0.2, 0.5), gridsize=(30, 70)), g["kde_fine"], rtol=1e-11)
    d, m = g["dens"], g["mask"]
    for name, kw in (("F_plain", dict(temperature=300.0)), ("F_mask", dict(temperature=310.0, mask=m)),
                     ("F_inpaint", dict(temperature=310.0, mask=m, inpaint=True)),
                     ("F_tiny", dict(temperature=300.0, tiny=0.01))):
        np.testing.assert_allclose(free_energy_from_density(d, **kw), g[name], rtol=1e-13, atol=1e-12, equal_nan=True)
    for name, data, kw in (("pmf_plain", g["cv"], dict(bins=60)), ("pmf_smooth", g["cv"], dict(bins=80, smoothing_sigma=1.5)),
                           ("pmf_periodic", g["tx"], dict(bins=36, periodic=True, range_=(-np.pi, np.pi), smoothing_sigma=0.8)),
                           ("pmf_range", g["cv"], dict(bins=25, range_=(-0.5, 2.0), temperature=350.0))):
        r = generate_1d_pmf(data, **kw)
        np.testing.assert_allclose(r.edges, g[f"{name}_edges"], rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(r.counts, g[f"{name}_counts"], rtol=1e-13)
        np.testing.assert_allclose(r.F, g[f"{name}_F"], rtol=1e-12, atol=1e-12)
        assert r.output_shape == (kw["bins"],)
    big = np.random.default_rng(0).uniform(-np.pi, np.pi, size=(2, 300_000))
    np.testing.assert_allclose(periodic_kde_2d(big[0], big[1], gridsize=(90, 33)),
                               npport.periodic_kde_2d(big[0][:300_000], big[1], gridsize=(90, 33)), rtol=1e-10)
    with pytest.raises(ValueError):
        periodic_kde_2d([], [])
    with pytest.raises(ValueError):
        generate_1d_pmf(g["cv"], bins=0)
    with pytest.raises(ValueError):
        free_energy_from_density(d, 0.0)


def test_order_statistics_radix_select(engine):
    """msm_order_statistics: exact k-th smallest by digit-wise histogram selection, against np.sort."""
    rng = np.random.default_rng(12)
    for n, d in ((1, 1), (7, 1), (100_003, 2), (1_000_000, 3)):
        X = rng.normal(size=(n, d)) * 10.0 ** rng.integers(-3, 4, size=d)
        X[rng.integers(0, n, size=max(1, n // 10))] = np.round(X[rng.integers(0, n, size=max(1, n // 10))], 1)   # ties
        if n > 10:
            X[:5, 0] = [0.0, -0.0, 1e-310, -1e-310, np.inf]
        xd = engine.to_device(np.ascontiguousarray(X))
        for col in range(d):
            ranks = np.unique(np.concatenate([[0, n - 1, n // 2, (n - 1) // 2], rng.integers(0, n, size=6)]))
            got = engine.order_statistics(xd, ranks, col=col)
            want = np.sort(X[:, col])[ranks]
            np.testing.assert_array_equal(got, want)
    with pytest.raises(ValueError):
        engine.order_statistics(engine.to_device(np.zeros((4, 1))), [4])


FES2D = {
    "adaptive": ("a", "b", dict(bins=(40, 40))),
    "fixed": ("a", "b", dict(bins=(30, 50), grid_strategy="fixed", min_count=2)),
    "ranges": ("a", "b", dict(bins=(25, 25), ranges=((-2.0, 2.5), (-2.0, 1.5)), grid_strategy="fixed", temperature=330.0)),
    "torus": ("phi", "psi", dict(bins=(36, 36), periodic=(True, True), grid_strategy="fixed")),
    "half_torus": ("phi", "b", dict(bins=(20, 20), periodic=(True, False))),
    "auto": ("a", "b", dict(bins=(40, 40), fes_smoothing_mode="auto")),
    "always": ("a", "b", dict(bins=(40, 40), grid_strategy="fixed", config={"fes_smoothing_mode": "always", "fes_h0": 0.9})),
}


@pytest.mark.parametrize("case", sorted(FES2D))
def test_generate_2d_fes_vs_reference_golden(golden, case):
    """generate_2d_fes (free_energy.py:417-868): grids from device order statistics (1 % / 99 % crop, FD rule),
    device histograms, host grid logic; golden made by importing the reference.  Counts are integers: the density,
    masks and grid shapes must match exactly up to the final divisions (1e-13); F to 1e-11."""
    from pmarlo_amd.markov_state_model.free_energy import generate_2d_fes

    g = golden("fes2d.npz")
    u, v, kw = FES2D[case]
    r = generate_2d_fes(g[u], g[v], **kw)
    md = r.metadata
    assert tuple(md["grid_shape"]) == tuple(g[f"{case}_shape"]) == r.output_shape
    np.testing.assert_allclose(r.xedges, g[f"{case}_xedges"], rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(r.yedges, g[f"{case}_yedges"], rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(md["counts"], g[f"{case}_density"], rtol=1e-13)
    np.testing.assert_array_equal(md["mask"], g[f"{case}_mask"])
    np.testing.assert_allclose(r.F, g[f"{case}_F"], rtol=1e-11, atol=1e-11, equal_nan=True)
    assert md["empty_bins_fraction"] == pytest.approx(float(g[f"{case}_empty"]), rel=1e-14)
    assert md["smoothing"]["applied_fraction"] == pytest.approx(float(g[f"{case}_applied"]), rel=1e-14)
    assert ("sparse_warning" in md) == (md["empty_bins_fraction"] > 0.5)
    assert r.temperature == kw.get("temperature", 300.0) and r.free_energy is r.F


def test_generate_2d_fes_errors():
    from pmarlo_amd.markov_state_model.free_energy import generate_2d_fes

    x = np.linspace(0, 1, 50)
    for bad in (dict(bins=(0, 5)), dict(temperature=0.0), dict(grid_strategy="magic"), dict(min_count=-1),
                dict(fes_smoothing_mode="sometimes"), dict(ranges=((0, 1),)), dict(ranges=((1.0, 0.0), (0.0, 1.0)))):
        with pytest.raises(ValueError):
            generate_2d_fes(x, x[::-1].copy(), **bad)
    with pytest.raises(ValueError):
        generate_2d_fes([], [])


def test_generate_2d_fes_crop_path_when_scipy_masks(golden, monkeypatch):
    """With a scipy whose mquantiles returns masked arrays the reference crops to the 1 % / 99 % quantiles and clips
    the samples; that branch (device order statistics + clip kernel) is exercised by switching the probe on."""
    from scipy.stats.mstats import mquantiles

    import pmarlo_amd.markov_state_model.free_energy as fe

    g = golden("fes2d.npz")
    a, b = g["a"], g["b"]
    monkeypatch.setattr(fe, "_reference_crop_is_live", lambda: True)
    r = fe.generate_2d_fes(a, b, bins=(30, 30), grid_strategy="fixed")
    qa, qb = mquantiles(a, prob=[0.01, 0.99]), mquantiles(b, prob=[0.01, 0.99])
    np.testing.assert_allclose(r.xedges[[0, -1]], qa, rtol=1e-15)
    np.testing.assert_allclose(r.yedges[[0, -1]], qb, rtol=1e-15)
    bx, by = r.metadata["grid_shape"]
    H, _, _ = np.histogram2d(np.clip(a, *qa), np.clip(b, *qb), bins=(np.linspace(*qa, bx + 1), np.linspace(*qb, by + 1)))
    np.testing.assert_allclose(r.metadata["counts"], H / (H.sum() * np.diff(r.xedges)[0] * np.diff(r.yedges)[0]), rtol=1e-13)
    assert H.sum() == a.size                                   # clipping keeps every sample on the grid


def test_fes_calculator_vs_reference_golden(golden):
    """FESCalculator.calculate_fes (MSM-reweighted FES in kT): device gather of pi[state], data range, weighted
    histogram; golden made by importing the reference.  Fixed-point weighted sums: 1e-11 on F."""
    from types import SimpleNamespace

    from pmarlo_amd.markov_state_model.free_energy import FESCalculator

    g = golden("fes_calculator.npz")
    proj, dtr = [g["proj0"], g["proj1"]], [g["d0"], g["d1"]]
    msm = SimpleNamespace(stationary_distribution=g["pi"])
    calc = FESCalculator({"temperature": 310.0})
    for name, kw in (("default", dict(bins=40)), ("dims", dict(bins=25, dim_x=2, dim_y=0, max_energy_cap_kt=None)),
                     ("cap", dict(bins=30, max_energy_cap_kt=3.0))):
        grid, F = calc.calculate_fes(proj, msm, dtrajs=dtr, **kw)
        np.testing.assert_allclose(grid[0], g[f"{name}_xx"], rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(grid[1], g[f"{name}_yy"], rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(F, g[f"{name}_F"], rtol=1e-11, atol=1e-11)
    assert calc.calculate_fes([], msm, dtrajs=dtr) == (None, None)
    assert calc.calculate_fes(proj, None, dtrajs=dtr) == (None, None)
    assert calc.calculate_fes(proj, msm, dtrajs=[dtr[0]]) == (None, None)              # frame counts differ
    assert calc.calculate_fes(proj, msm, dtrajs=dtr, dim_x=5) == (None, None)
    with_attr = SimpleNamespace(stationary_distribution=g["pi"], discrete_trajectories=dtr)
    assert calc.calculate_fes(proj, with_attr, bins=10)[1].shape == (10, 10)


def test_msm_reweighted_fes_vs_reference_mixin_golden(golden):
    """markov_state_model/fes.py against FESMixin's methods (golden made by calling them on a stand-in object):
    frame weights pi[state], bin choice, weighted density histogram with wrap / reflect smoothing, free energy."""
    from pmarlo_amd.markov_state_model.fes import (choose_bins, generate_free_energy_surface, histogram_to_free_energy,
                                                    stationary_frame_weights, weighted_density_histogram)

    g = golden("msm_fes.npz")
    dtr = [g["d0"], g["d1"]]
    w = stationary_frame_weights(dtr, g["pi"])
    np.testing.assert_array_equal(w, g["weights"])
    assert [choose_bins(t, b) for t, b in ((0, 30), (7500, 50), (7500, 44), (10 ** 6, 10), (90000, 58))] == g["bins"].tolist()
    for name, a, b, ranges, per in (("torsion", g["phi"], g["psi"], [(-180.0, 180.0), (-180.0, 180.0)], True),
                                    ("plain", g["u"], g["v"], None, False)):
        H, xe, ye = weighted_density_histogram(a, b, w, choose_bins(w.size, 50), ranges, smooth_sigma=0.6, periodic=per)
        np.testing.assert_allclose(xe, g[f"{name}_xe"], rtol=1e-14, atol=1e-13)
        np.testing.assert_allclose(ye, g[f"{name}_ye"], rtol=1e-14, atol=1e-13)
        np.testing.assert_allclose(H, g[f"{name}_H"], rtol=1e-11, atol=1e-300)
        np.testing.assert_allclose(histogram_to_free_energy(H, 300.0), g[f"{name}_F"], rtol=1e-10, atol=1e-10)
    res = generate_free_energy_surface(g["phi"], g["psi"], dtr, g["pi"])
    np.testing.assert_allclose(res["free_energy"], g["torsion_F"], rtol=1e-10, atol=1e-10)
    assert res["cv1_name"] == "phi" and res["xedges"][0] == -180.0 and res["temperature"] == 300.0
    with pytest.raises(ValueError, match="too sparse"):
        histogram_to_free_energy(np.zeros((3, 3)), 300.0)
    with pytest.raises(ValueError, match="Could not generate histogram"):
        weighted_density_histogram([1.0], [1.0, 2.0], [1.0], 40)


def test_output_whitening_vs_reference_golden(golden):
    """apply_whitening_from_metadata / ensure_fes_inputs_whitened: device projection + moment passes with the
    d x d algebra on the host; golden made by importing the reference (1e-11: different summation order)."""
    from pmarlo_amd.analysis.fes import compute_weighted_fes
    from pmarlo_amd.analysis.project_cv import apply_whitening_from_metadata

    g = golden("whitening.npz")
    Y, mean, W = g["Y"], g["mean"], g["W"]
    md = {"output_mean": mean.tolist(), "output_transform": W.tolist(), "output_transform_applied": "false"}
    out, applied = apply_whitening_from_metadata(Y, md)
    assert applied == bool(g["applied"]) and md["output_transform_applied"] is True
    np.testing.assert_allclose(out, g["whitened"], rtol=0, atol=1e-11)
    # (the reference multiplies by L^-1 rather than L^-T, so the batch covariance comes out near, not at, identity)
    again, did = apply_whitening_from_metadata(Y, md)
    assert did == bool(g["again_applied"]) and again is not None
    np.testing.assert_array_equal(again, g["again"])
    few, _ = apply_whitening_from_metadata(Y[:2], {"output_mean": mean, "output_transform": W})
    np.testing.assert_allclose(few, g["few"], rtol=0, atol=1e-12)
    ds = {"X": Y.copy(), "splits": {"train": {"X": Y[:3000].copy()}, "val": {"X": Y[3000:].copy()}},
          "__artifacts__": {"mlcv_deeptica": {"output_mean": mean.tolist(), "output_transform": W.tolist()}}}
    res = compute_weighted_fes(ds, split="train", bins=14, method="kde")
    np.testing.assert_allclose(ds["X"], g["ds_X"], atol=1e-11)
    np.testing.assert_allclose(ds["splits"]["train"]["X"], g["ds_train"], atol=1e-11)
    np.testing.assert_allclose(ds["splits"]["val"]["X"], g["ds_val"], atol=1e-11)
    np.testing.assert_allclose(res["xedges"], g["fes_xedges"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(res["histogram"], g["fes_hist"], rtol=1e-8)
    np.testing.assert_allclose(res["free_energy"], g["fes_F"], atol=1e-7)
    for bad in ({"output_mean": mean}, None):
        with pytest.raises((ValueError, TypeError)):
            apply_whitening_from_metadata(Y, bad)
    with pytest.raises(ValueError, match="boolean-like"):
        apply_whitening_from_metadata(Y, {"output_mean": mean, "output_transform": W, "output_transform_applied": "maybe"})


def test_ensure_msm_inputs_whitened(golden):
    """analysis/msm.ensure_msm_inputs_whitened: top-level X whitened once from the DeepTICA metadata."""
    from pmarlo_amd.analysis.msm import ensure_msm_inputs_whitened

    g = golden("whitening.npz")
    ds = {"X": g["Y"].copy(), "__artifacts__": {"mlcv_deeptica": {"output_mean": g["mean"].tolist(),
                                                                   "output_transform": g["W"].tolist()}}}
    assert ensure_msm_inputs_whitened(ds) is True
    np.testing.assert_allclose(ds["X"], g["whitened"], atol=1e-11)
    before = ds["X"].copy()
    assert ensure_msm_inputs_whitened(ds) is False                    # flag set: not applied twice
    np.testing.assert_array_equal(ds["X"], before)
    assert ensure_msm_inputs_whitened({"X": g["Y"]}) is False and ensure_msm_inputs_whitened([1, 2]) is False
    assert ensure_msm_inputs_whitened({"X": g["Y"], "__artifacts__": {"mlcv_deeptica": {"output_mean": None}}}) is False
