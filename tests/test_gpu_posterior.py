"""Posterior transition-matrix samples and the Bayesian implied-timescale band
(msm_sample_transition_matrices + batched msm_spectrum).

deeptime's BayesianMSM stream cannot be reproduced (parity unpinned: third-party sampler absent), so
the checks are: the generator against Random123's known answers; the device variates against the
numpy restatement of the same algorithm (oracle/npport.py, 1e-9: libm differences only); the
Dirichlet law itself (moments, Kolmogorov-Smirnov against the Beta marginal); independence from the
launch geometry; and the per-sample spectra / median / percentile band against numpy's eig on the very
matrices the device drew (1e-6 relative, north star 1e-5)."""
import numpy as np
import pytest
from scipy import stats

from oracle import npport
from tests import _gen
from pmarlo_amd.markov_state_model import compute_implied_timescales

pytestmark = pytest.mark.gpu


def _estimate(engine, C, alpha):
    from pmarlo_amd._lib import check, lib

    k = C.shape[0]
    cd = engine.to_device(C.astype(np.int64))
    T, act, inv = engine.empty((k, k), np.float64), engine.empty((k,), np.int32), engine.empty((k,), np.int32)
    na, rows = engine.empty((1,), np.int32), engine.empty((k,), np.float64)
    check(lib.msm_transition_matrix(engine.handle, cd.ptr, 0, k, 1, float(alpha), 1e-12, T.ptr, act.ptr, inv.ptr,
                                    na.ptr, rows.ptr, None), engine.handle)
    return cd, act, na


def test_philox_known_answers(engine):
    kat = [(0, [0, 0, 0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
           ((0xFFFFFFFF << 32) | 0xFFFFFFFF, [0xFFFFFFFF] * 4, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
           ((0x299F31D0 << 32) | 0xA4093822, [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344],
            [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1])]
    for key, ctr, want in kat:
        np.testing.assert_array_equal(engine.philox4x32(key, ctr), np.asarray(want, np.uint32))


def test_samples_match_the_numpy_restatement(engine):
    rng = np.random.default_rng(4)
    k = 9
    C = rng.integers(0, 60, size=(k, k))
    C[rng.random((k, k)) < 0.4] = 0
    C[3, :] = 0
    C[:, 3] = 0                     # state 3 never visited: outside the active set
    C[5, :] = 0                     # state 5 only entered: its row is prior only (shape 1e-3 everywhere)
    cd, act, na = _estimate(engine, C, 1e-3)
    n = int(na.to_host()[0])
    active = act.to_host()[:n]
    assert n == k - 1 and 3 not in active
    got = engine.sample_transition_matrices(cd, act, na, alpha=1e-3, seed=99, n_samples=40, first_sample=5).to_host()
    want = npport.sample_transition_matrices(C[np.ix_(active, active)] + 1e-3, 99, 5, 40)
    np.testing.assert_allclose(got[:, :n, :n], want, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(got[:, :n, :n].sum(-1), 1.0, rtol=1e-14)
    assert np.all(got[:, :n, :n] >= 0)


def test_samples_do_not_depend_on_the_batch(engine):
    C = np.random.default_rng(1).integers(1, 30, size=(300, 300))      # rows longer than one workgroup pass
    cd, act, na = _estimate(engine, C, 1e-3)
    whole = engine.sample_transition_matrices(cd, act, na, alpha=1e-3, seed=3, n_samples=6).to_host()
    tail = engine.sample_transition_matrices(cd, act, na, alpha=1e-3, seed=3, n_samples=2, first_sample=4).to_host()
    np.testing.assert_array_equal(whole[4:], tail)
    other = engine.sample_transition_matrices(cd, act, na, alpha=1e-3, seed=4, n_samples=1).to_host()
    assert np.abs(other[0] - whole[0]).max() > 1e-3


def test_dirichlet_law(engine):
    C = np.array([[120, 14, 3, 0], [9, 60, 0, 2], [1, 0, 25, 6], [0, 3, 4, 30]])
    alpha = 0.5
    cd, act, na = _estimate(engine, C, alpha)
    S = 6000
    T = engine.sample_transition_matrices(cd, act, na, alpha=alpha, seed=11, n_samples=S).to_host()
    A = C + alpha
    A0 = A.sum(1, keepdims=True)
    mean, var = A / A0, A * (A0 - A) / (A0 ** 2 * (A0 + 1))
    assert np.all(np.abs(T.mean(0) - mean) < 5.0 * np.sqrt(var / S))
    assert np.all(np.abs(T.var(0) - var) < 0.15 * var + 1e-12)
    for i, j in ((0, 1), (1, 0), (2, 3), (3, 0)):          # marginal of a Dirichlet cell: Beta(a_ij, A_i - a_ij)
        p = stats.kstest(T[:, i, j], stats.beta(A[i, j], A0[i, 0] - A[i, j]).cdf).pvalue
        assert p > 1e-4, (i, j, p)
    # rows are independent: correlation between cells of different rows ~ 0
    assert abs(np.corrcoef(T[:, 0, 1], T[:, 1, 0])[0, 1]) < 5.0 / np.sqrt(S)


def test_sampled_spectra_match_numpy_on_the_same_matrices(engine):
    rng = np.random.default_rng(8)
    k, lag = 40, 3
    # metastable chain: 4 blocks of 10 states
    P = np.full((k, k), 0.002)
    for b in range(4):
        P[10 * b:10 * b + 10, 10 * b:10 * b + 10] += rng.uniform(0.05, 0.15, size=(10, 10))
    P /= P.sum(1, keepdims=True)
    x = np.zeros(60_000, dtype=np.int64)
    cdf = np.cumsum(P, axis=1)
    u = rng.random(x.size)
    for t in range(1, x.size):
        x[t] = min(k - 1, int(np.searchsorted(cdf[x[t - 1]], u[t])))
    C = np.zeros((k, k), dtype=np.int64)
    np.add.at(C, (x[:-lag], x[lag:]), 1)
    cd, act, na = _estimate(engine, C, 1e-3)
    S = 64
    Td = engine.sample_transition_matrices(cd, act, na, alpha=1e-3, seed=5, n_samples=S)
    spec = engine.spectrum(Td, n=engine.to_device(np.full(S, k, np.int32)), n_its=3, lags=np.full(S, float(lag)),
                           want_pi=False)
    Th = Td.to_host()
    want = npport.its_posterior_summary(Th, lag, 3, 2.5, 97.5)
    per_sample = np.stack([npport.its_from_transition_matrix(T, lag, 3)[1] for T in Th])
    np.testing.assert_allclose(spec["its_ts"], per_sample, rtol=1e-6)
    np.testing.assert_allclose(np.nanmedian(spec["its_ts"], axis=0), want["timescales"], rtol=1e-6)
    np.testing.assert_allclose(np.nanpercentile(spec["its_ts"], 97.5, axis=0), want["timescales_ci"][:, 1], rtol=1e-6)


def test_bayesian_its_two_state(engine):
    rng = np.random.default_rng(0)
    n = 40_000
    traj = (np.cumsum(rng.random(n) < 0.1) % 2).astype(int)        # lambda2 = 0.8
    t_true = -1.0 / np.log(0.8)
    lags = [1, 2, 3, 5]
    det = compute_implied_timescales([traj], 2, lag_times=lags, n_timescales=1, n_samples=0)
    res = compute_implied_timescales([traj], 2, lag_times=lags, n_timescales=1, n_samples=200, ci=0.95,
                                     random_state=7, return_samples=True)
    again = compute_implied_timescales([traj], 2, lag_times=lags, n_timescales=1, n_samples=200, random_state=7)
    np.testing.assert_array_equal(res.timescales, again.timescales)            # seeded: reproducible
    np.testing.assert_array_equal(res.timescales_ci, again.timescales_ci)
    assert res.timescales.shape == (4, 1) and res.timescales_ci.shape == (4, 1, 2)
    lo, hi = res.timescales_ci[:, 0, 0], res.timescales_ci[:, 0, 1]
    assert np.all(lo < res.timescales[:, 0]) and np.all(res.timescales[:, 0] < hi)
    assert np.all(np.abs(res.timescales - det.timescales) / det.timescales < 0.03)   # median ~ point estimate
    assert np.all(lo < t_true * 1.05) and np.all(hi > t_true * 0.95)
    # the band narrows as 1/sqrt(data): a 4x longer trajectory roughly halves it
    long = (np.cumsum(np.random.default_rng(1).random(4 * n) < 0.1) % 2).astype(int)
    wide = compute_implied_timescales([long], 2, lag_times=[1], n_timescales=1, n_samples=200, random_state=7)
    ratio = (hi[0] - lo[0]) / (wide.timescales_ci[0, 0, 1] - wide.timescales_ci[0, 0, 0])
    assert 1.4 < ratio < 2.9
    # analytic check of the band at lag 1: lambda2 = 1 - p01 - p10 with independent Beta rows
    C = np.zeros((2, 2))
    np.add.at(C, (traj[:-1], traj[1:]), 1)
    a = C + 1e-3
    draws = 1.0 - stats.beta(a[0, 1], a[0, 0]).rvs(200_000, random_state=1) - stats.beta(a[1, 0], a[1, 1]).rvs(
        200_000, random_state=2)
    want = np.percentile(-1.0 / np.log(draws), [2.5, 97.5])
    np.testing.assert_allclose([lo[0], hi[0]], want, rtol=0.02)
    np.testing.assert_allclose(res.rates, np.nanmedian(1.0 / res.samples["timescales"], axis=1))
    win = compute_implied_timescales([traj], 2, lag_times=lags, n_timescales=1, n_samples=0, plateau_m=3,
                                     plateau_epsilon=0.2, time_per_frame_ps=2.0).recommended_lag_window
    assert win == (2.0, 10.0)


def test_deterministic_its_from_counts(engine):
    """Fall-back estimate: spectrum of rownorm((C + C') / 2) against the numpy restatement of the intended form."""
    from pmarlo_amd.markov_state_model import deterministic_its_from_counts

    rng = np.random.default_rng(2)
    C = rng.poisson(2.0, size=(30, 30)).astype(float)
    for b in range(3):
        C[10 * b:10 * b + 10, 10 * b:10 * b + 10] += rng.poisson(40.0, size=(10, 10))
    ev, ts, rates = deterministic_its_from_counts(C, 4, 3)
    ev_ref, ts_ref = npport.reversible_its_from_counts(C, 4, 3)
    np.testing.assert_allclose(ev, ev_ref, rtol=1e-9)
    np.testing.assert_allclose(ts, ts_ref, rtol=1e-8)
    np.testing.assert_allclose(rates, 1.0 / ts_ref, rtol=1e-8)
    ev2, ts2, _ = deterministic_its_from_counts(C[:2, :2], 1, 4)          # more timescales than the chain has
    assert np.isfinite(ts2[0]) and np.all(np.isnan(ts2[1:])) and np.all(ev2[1:] == 0.0)


def test_deterministic_its_reference_quirk(engine):
    """reference_quirk=True: what the reference's fall-back RETURNS (uniform "pi" -> eigvalsh on the lower triangle
    of the non-symmetric T, S/markov_state_model/_its.py:753-787) against its numpy restatement, and how far that is
    from the intended estimate on a metastable chain."""
    from pmarlo_amd.markov_state_model import deterministic_its_from_counts

    lab = _gen.metastable_labels(200_000, 60, 4, seed=2)
    C = np.zeros((60, 60))
    np.add.at(C, (lab[:-5], lab[5:]), 1.0)
    ev_q, ts_q, rates_q = deterministic_its_from_counts(C, 5, 4, reference_quirk=True)
    ev_r, ts_r, rates_r = npport.reference_quirk_its_from_counts(C, 5, 4)
    np.testing.assert_allclose(ev_q, ev_r, rtol=1e-9)
    np.testing.assert_allclose(ts_q, ts_r, rtol=1e-8)
    np.testing.assert_allclose(rates_q, rates_r, rtol=1e-8)
    ev_i, ts_i, _ = deterministic_its_from_counts(C, 5, 4)
    # the quirk is not a rounding matter: it moves the slowest timescale of this chain by about 18 %
    rel = np.abs(ts_q - ts_i) / ts_i
    assert 0.05 < rel[0] < 0.4 and rel[3] < 1e-2
    # symmetric counts with equal row sums: T is symmetric, both forms agree
    S = np.full((6, 6), 2.0) + 10.0 * np.eye(6)
    a = deterministic_its_from_counts(S, 2, 3, reference_quirk=True)
    b = deterministic_its_from_counts(S, 2, 3)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-8)
    # more timescales than states: padded as the reference pads
    e2, t2, _ = deterministic_its_from_counts(C[:3, :3], 1, 5, reference_quirk=True)
    assert np.all(np.isnan(t2[2:])) and np.all(e2[2:] == 0.0)


def test_its_input_rules_of_the_reference(engine):
    """_validate_its_inputs (S/markov_state_model/_its.py:453-524): an empty trajectory in the list empties the result
    (max_valid_lag = -1), lags above min(len) - 1 are dropped, effective_frames caps the largest lag with a ValueError."""
    traj = _gen.metastable_labels(4000, 6, 2, seed=1)
    res = compute_implied_timescales([traj, np.array([], dtype=np.int32)], 6, lag_times=[1, 2], n_timescales=2, n_samples=0)
    assert res.lag_times.size == 0 and res.timescales.shape == (0, 2)
    assert compute_implied_timescales([], 6, lag_times=[1], n_timescales=2, n_samples=0).lag_times.size == 0
    short = compute_implied_timescales([traj, traj[:4]], 6, lag_times=[1, 3, 4, 50], n_timescales=2, n_samples=0)
    np.testing.assert_array_equal(short.lag_times, [1, 3])
    with pytest.raises(ValueError, match="effective frames"):
        compute_implied_timescales([traj], 6, lag_times=[1, 100], n_timescales=2, n_samples=0, effective_frames=100)
    ok = compute_implied_timescales([traj], 6, lag_times=[1, 99], n_timescales=2, n_samples=0, effective_frames=100)
    np.testing.assert_array_equal(ok.lag_times, [1, 99])
