"""Dense solves on T (msm_solve_f64, msm_reactive_flux, msm_lump_macro, msm_macro_mfpt) against
the numpy oracle and analytic invariants.  Tolerances 1e-10 relative (LU with partial pivoting on
both sides, different elimination order)."""
import numpy as np
import pytest

from oracle import npport
from pmarlo_amd.markov_state_model.tpt import (compute_committor, compute_macro_mfpt, compute_macro_populations,
                                               lump_micro_to_macro_T, reactive_flux)

pytestmark = pytest.mark.gpu


def _metastable_T(n, seed, n_macro=4, reversible=False):
    rng = np.random.default_rng(seed)
    C = rng.random((n, n)) * 0.02
    per = n // n_macro
    for b in range(n_macro):
        s = slice(b * per, (b + 1) * per if b < n_macro - 1 else n)
        C[s, s] += rng.random((C[s, s].shape)) + 0.2
    if reversible:
        C = C + C.T
    return C / C.sum(axis=1, keepdims=True)


@pytest.mark.parametrize("n,nrhs", [(1, 1), (5, 2), (64, 3), (300, 1), (700, 4)])
def test_solve(engine, n, nrhs):
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)) + (0.0 if n > 1 else 2.0)
    A[0, 0] = 0.0 if n > 1 else A[0, 0]          # forces a row exchange
    B = rng.normal(size=(n, nrhs))
    Ad, Bd = engine.to_device(A), engine.to_device(B)
    assert engine.solve(Ad, Bd) == 0
    X = Bd.to_host()
    np.testing.assert_allclose(A @ X, B, atol=1e-9 * max(1.0, np.abs(B).max()) * n)
    np.testing.assert_allclose(X, np.linalg.solve(A, B), rtol=1e-7, atol=1e-9)
    S = np.ones((4, 4))
    assert engine.solve(engine.to_device(S), engine.to_device(np.ones((4, 1)))) != 0   # singular is reported


@pytest.mark.parametrize("n,reversible", [(12, False), (100, True), (500, False)])
def test_reactive_flux_vs_oracle_and_invariants(n, reversible):
    T = _metastable_T(n, seed=n, reversible=reversible)
    pi = npport.stationary_distribution(T)
    A, B = [0, 1, 2], [n - 1, n - 2]
    got = reactive_flux(T, pi, A, B)
    want = npport.reactive_flux(T, pi, A, B)
    np.testing.assert_allclose(got.forward_committor, want["qplus"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got.backward_committor, want["qminus"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got.gross_flux, want["gross"], rtol=1e-9, atol=1e-18)
    np.testing.assert_allclose(got.net_flux, want["net"], rtol=1e-8, atol=1e-16)
    np.testing.assert_allclose([got.total_flux, got.rate, got.mfpt], [want["total_flux"], want["rate"], want["mfpt"]],
                               rtol=1e-9)
    # invariants: boundary values, probabilities, flux conservation at intermediate states
    qp, qm = got.forward_committor, got.backward_committor
    assert np.all(qp[A] == 0) and np.all(qp[B] == 1) and np.all(qm[A] == 1) and np.all(qm[B] == 0)
    assert qp.min() >= -1e-12 and qp.max() <= 1 + 1e-12
    inter = np.setdiff1d(np.arange(n), A + B)
    np.testing.assert_allclose(got.gross_flux[inter].sum(axis=1), got.gross_flux[:, inter].sum(axis=0), rtol=1e-7,
                               atol=1e-15)
    if reversible:
        np.testing.assert_allclose(qm, 1.0 - qp, atol=1e-9)
    np.testing.assert_allclose(compute_committor(T, A, B), want["qplus"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(compute_committor(T, A, B, forward=False, stationary_distribution=pi), want["qminus"],
                               rtol=1e-9, atol=1e-12)


def test_backward_committor_without_pi_uses_device_stationary_vector():
    T = _metastable_T(40, seed=2)
    q = compute_committor(T, [0], [39], forward=False)
    np.testing.assert_allclose(q, npport.committor(T, [0], [39], forward=False), rtol=1e-7, atol=1e-10)


def test_tpt_errors():
    T = _metastable_T(10, seed=1)
    pi = npport.stationary_distribution(T)
    with pytest.raises(ValueError):
        reactive_flux(T, pi, [0, 1], [1, 2])
    with pytest.raises(ValueError):
        reactive_flux(None, pi, [0], [1])
    with pytest.raises(ValueError):
        compute_committor(T, [0], [10])


@pytest.mark.parametrize("n,n_macro", [(30, 3), (500, 6), (64, 64)])
def test_lumping_and_macro_mfpt(n, n_macro):
    rng = np.random.default_rng(n)
    T = _metastable_T(n, seed=n + 1, n_macro=min(n_macro, 6))
    pi = npport.stationary_distribution(T)
    macro = np.arange(n) % n_macro if n_macro == n else np.sort(rng.integers(0, n_macro, n))
    macro[:n_macro] = np.arange(n_macro)           # every macrostate populated
    Tm = lump_micro_to_macro_T(T, pi, macro)
    np.testing.assert_allclose(Tm, npport.lump_micro_to_macro_T(T, pi, macro), rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(Tm.sum(axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(compute_macro_populations(pi, macro), npport.macro_populations(pi, macro), rtol=1e-12)
    if n_macro <= 8:
        M = compute_macro_mfpt(Tm)
        np.testing.assert_allclose(M, npport.macro_mfpt(Tm), rtol=1e-9)
        assert np.all(np.diag(M) == 0)
    # two-state analytic: mfpt 0 -> 1 = 1 / p01
    M2 = compute_macro_mfpt(np.array([[0.9, 0.1], [0.25, 0.75]]))
    np.testing.assert_allclose(M2, [[0.0, 10.0], [4.0, 0.0]], rtol=1e-12)


def _two_route_network():
    """0 -> {1 -> 2 | 3} -> 4: a strong route over 1, 2 and a weak one over 3 (symmetric weights: reversible)."""
    W = np.zeros((5, 5))
    for i, j, w in ((0, 1, 4.0), (1, 2, 3.0), (2, 4, 4.0), (0, 3, 1.0), (3, 4, 1.0), (1, 3, 0.2)):
        W[i, j] = W[j, i] = w
    W += np.diag([6.0, 5.0, 5.0, 4.0, 6.0])
    T = W / W.sum(1, keepdims=True)
    pi = W.sum(1) / W.sum()
    return T, pi


def test_pathway_decomposition():
    from pmarlo_amd.markov_state_model import pathway_decomposition, reactive_flux

    T, pi = _two_route_network()
    flux = reactive_flux(T, pi, [0], [4])
    paths, caps = pathway_decomposition(T, pi, [0], [4], fraction=1.0)
    assert paths[0] == [0, 1, 2, 4] and paths[1] == [0, 3, 4]               # widest bottleneck first
    assert all(p[0] == 0 and p[-1] == 4 for p in paths)
    assert np.all(np.diff(caps) <= 1e-15)                                     # capacities never grow
    np.testing.assert_allclose(caps.sum(), flux.total_flux, rtol=1e-12)       # the paths carry the whole flux
    for p, c in zip(paths, caps):                                             # no path exceeds any of its edges
        assert all(flux.net_flux[a, b] >= c * (1 - 1e-12) for a, b in zip(p[:-1], p[1:]))
    few, fc = pathway_decomposition(T, pi, [0], [4], fraction=0.5)
    assert len(few) == 1 and fc[0] >= 0.5 * flux.total_flux
    one, _ = pathway_decomposition(T, pi, [0], [4], fraction=1.0, maxiter=1)
    assert len(one) == 1
    # a larger metastable chain: conservation again, several source / sink states
    Tm = _metastable_T(40, seed=5, reversible=True)
    w, v = np.linalg.eig(Tm.T)
    pim = np.real(v[:, np.argmax(np.real(w))])
    pim /= pim.sum()
    fm = reactive_flux(Tm, pim, [0, 1, 2], [37, 38, 39])
    pm, cm = pathway_decomposition(Tm, pim, [0, 1, 2], [37, 38, 39], fraction=0.9)
    assert 0.9 * fm.total_flux * (1 - 1e-12) <= cm.sum() <= fm.total_flux * (1 + 1e-12)
    assert all(p[0] in (0, 1, 2) and p[-1] in (37, 38, 39) for p in pm)
    with pytest.raises(ValueError):
        pathway_decomposition(T, pi, [0], [4], fraction=0.0)


def test_coarse_grain_flux_tse_and_bottlenecks():
    from pmarlo_amd.markov_state_model import (coarse_grain_flux, compute_committor, find_bottleneck_states,
                                              identify_transition_state_ensemble, reactive_flux)

    T = _metastable_T(40, seed=7, reversible=True)
    w, v = np.linalg.eig(T.T)
    pi = np.real(v[:, np.argmax(np.real(w))])
    pi /= pi.sum()
    A, B = [0, 1, 2, 3], [36, 37, 38, 39]
    flux = reactive_flux(T, pi, A, B)
    sets = [list(range(0, 10)), list(range(10, 20)), list(range(20, 30))]      # 30..39 not named: one more set
    tpt_sets, cg = coarse_grain_flux(T, pi, A, B, sets)
    # source parts, intermediate parts, sink parts, in that order
    assert tpt_sets == [set(range(0, 4)), set(range(4, 10)), set(range(10, 20)), set(range(20, 30)), set(range(30, 36)),
                        set(range(36, 40))]
    assert cg.source_states == [0] and cg.sink_states == [5]
    S = np.zeros((40, 6))
    for q, s in enumerate(tpt_sets):
        S[sorted(s), q] = 1.0
    want = S.T @ flux.gross_flux @ S
    np.fill_diagonal(want, 0.0)
    np.testing.assert_allclose(cg.gross_flux, want, rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(cg.total_flux, flux.total_flux, rtol=1e-10)       # A and B are whole sets: conserved
    np.testing.assert_allclose(cg.rate, flux.rate, rtol=1e-10)
    np.testing.assert_allclose(cg.stationary_distribution, S.T @ pi, rtol=1e-13)
    assert cg.forward_committor[0] == 0.0 and cg.forward_committor[5] == pytest.approx(1.0, abs=1e-12)
    assert np.all((cg.forward_committor >= 0) & (cg.forward_committor <= 1 + 1e-12))
    np.testing.assert_allclose(cg.forward_committor, (S.T @ (pi * flux.forward_committor)) / (S.T @ pi), rtol=1e-13)
    with pytest.raises(ValueError, match="disjoint"):
        coarse_grain_flux(T, pi, A, B, [[0, 1], [1, 2]])
    q = compute_committor(T, A, B)
    np.testing.assert_array_equal(identify_transition_state_ensemble(T, A, B, tolerance=0.2),
                                  np.where((q >= 0.3) & (q <= 0.7))[0])
    through = 0.5 * (flux.gross_flux.sum(1) + flux.gross_flux.sum(0))
    np.testing.assert_array_equal(find_bottleneck_states(T, pi, A, B, top_n=5), np.argsort(through)[::-1][:5])
