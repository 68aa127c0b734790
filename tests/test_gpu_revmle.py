"""Reversible maximum-likelihood estimator on the GPU (msm_reversible_mle) against its numpy
restatement (oracle/npport.reversible_mle) and the properties that define it.

deeptime's MaximumLikelihoodMSM(reversible=True) is absent (parity unpinned); both sides iterate the
published fixed point, so at a tight stopping rule they agree to 1e-9 whatever the iteration count."""
import time

import numpy as np
import pytest

from oracle import npport
from pmarlo_amd.markov_state_model import build_simple_msm, fit_reversible_msm

pytestmark = pytest.mark.gpu


def _counts(k, seed, blocks=4):
    rng = np.random.default_rng(seed)
    C = rng.poisson(0.3, size=(k, k)).astype(float)
    w = max(1, k // blocks)
    for b in range(blocks):
        s = slice(b * w, (b + 1) * w if b < blocks - 1 else k)
        C[s, s] += rng.poisson(6.0, size=C[s, s].shape)
    return C + 1e-3


@pytest.mark.parametrize("k", [2, 7, 64, 300])
def test_reversible_mle_vs_oracle(engine, k):
    C = _counts(k, k)
    out = engine.reversible_mle(engine.to_device(C), maxerr=1e-13)
    T, pi = out["T"].to_host(), out["pi"].to_host()
    T_ref, pi_ref, _ = npport.reversible_mle(C, maxerr=1e-13)
    np.testing.assert_allclose(T, T_ref, rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(pi, pi_ref, rtol=1e-9)
    np.testing.assert_allclose(T.sum(1), 1.0, rtol=1e-14)
    flux = pi[:, None] * T
    np.testing.assert_allclose(flux, flux.T, rtol=1e-9, atol=1e-18)          # detailed balance
    np.testing.assert_allclose(pi @ T, pi, rtol=1e-9)
    # likelihood: not below the row-normalised (non-reversible) estimate's reversibilised neighbours
    ll = np.sum(C * np.log(T))
    Tsym = (C + C.T) / (C + C.T).sum(1, keepdims=True)
    assert ll >= np.sum(C * np.log(Tsym)) - 1e-9


def test_symmetric_counts_are_their_own_fixed_point(engine):
    C = _counts(20, 3)
    C = C + C.T
    out = engine.reversible_mle(engine.to_device(C))
    np.testing.assert_allclose(out["T"].to_host(), C / C.sum(1, keepdims=True), rtol=1e-12)
    np.testing.assert_allclose(out["pi"].to_host(), C.sum(1) / C.sum(), rtol=1e-12)
    assert out["iterations"] <= 32


def test_default_stopping_rule_and_cap(engine):
    C = _counts(40, 9)
    out = engine.reversible_mle(engine.to_device(C))
    assert out["err"] <= 1e-8 and out["iterations"] >= 32
    T_ref, _, it_ref = npport.reversible_mle(C)
    assert out["iterations"] >= it_ref - 1               # checked in blocks: never stops early
    np.testing.assert_allclose(out["T"].to_host(), T_ref, rtol=1e-5)
    capped = engine.reversible_mle(engine.to_device(C), maxerr=1e-300, maxiter=50)
    assert capped["iterations"] == 50


def test_build_simple_msm(engine):
    rng = np.random.default_rng(5)
    P = np.array([[0.9, 0.1, 0.0], [0.05, 0.9, 0.05], [0.0, 0.2, 0.8]])
    x = np.zeros(100_000, dtype=int)
    u = rng.random(x.size)
    cdf = np.cumsum(P, axis=1)
    for t in range(1, x.size):
        x[t] = int(np.searchsorted(cdf[x[t - 1]], u[t]))
    traj = np.where(np.arange(x.size) % 5000 == 0, -1, x)        # unassigned frames split the trajectory
    T, pi = build_simple_msm([traj, x[:3000] + 0], n_states=5, lag=1)     # states 3, 4 never visited
    assert T.shape == (5, 5) and pi.shape == (5,)
    np.testing.assert_array_equal(T[3:, 3:], np.eye(2))
    np.testing.assert_array_equal(pi[3:], 0.0)
    np.testing.assert_allclose(T[:3, :3], P, atol=0.01)
    w, v = np.linalg.eig(P.T)
    st = np.real(v[:, np.argmax(np.real(w))])
    np.testing.assert_allclose(pi[:3], st / st.sum(), atol=0.01)
    Tn, pin = build_simple_msm([x], lag=1)                        # n_states inferred
    assert Tn.shape == (3, 3)
    assert build_simple_msm([])[0].shape == (0, 0)
    T2, pi2, active = fit_reversible_msm(np.zeros((4, 4)))
    np.testing.assert_array_equal(T2, np.eye(4))
    assert active.size == 0


def test_large_matrix_timing(engine):
    C = _counts(2000, 1, blocks=8)
    d = engine.to_device(C)
    engine.reversible_mle(d, maxiter=64, maxerr=1e-300)
    engine.sync()
    t0 = time.perf_counter()
    out = engine.reversible_mle(d, maxiter=2048, maxerr=1e-300)
    engine.sync()
    per = (time.perf_counter() - t0) / out["iterations"]
    print(f"\\nreversible MLE k=2000: {per * 1e6:.1f} us / iteration ({out['iterations']} iterations)")
    T = out["T"].to_host()
    np.testing.assert_allclose(T.sum(1), 1.0, rtol=1e-13)
    assert per < 2e-3


def test_check_transition_matrix(engine):
    from pmarlo_amd.markov_state_model import check_transition_matrix

    C = _counts(12, 4)
    T, pi, _ = fit_reversible_msm(C)
    check_transition_matrix(T, pi)                                   # a valid pair passes silently
    check_transition_matrix(T, 5.0 * pi)                             # pi is normalised first
    with pytest.raises(ValueError, match="invariance"):
        check_transition_matrix(T, np.roll(pi, 1))
    bad = T.copy()
    bad[0, 0] -= 0.2
    bad[0, 1] += 0.1
    with pytest.raises(ValueError, match="stochasticity"):
        check_transition_matrix(bad, pi)
    neg = T.copy()
    neg[0, 0], neg[0, 1] = -0.1, T[0, 1] + T[0, 0] + 0.1
    with pytest.raises(ValueError, match="Negative"):
        check_transition_matrix(neg, pi)
    with pytest.raises(ValueError, match="size mismatch"):
        check_transition_matrix(T, pi[:-1])
    with pytest.raises(ValueError, match="normalisable"):
        check_transition_matrix(T, np.zeros_like(pi))
    # two closed blocks: a vector supported on one block is stationary; the other block is unreachable from it
    R = np.zeros((6, 6))
    R[:3, :3] = T[:3, :3] / T[:3, :3].sum(1, keepdims=True)
    R[3:, 3:] = T[3:6, 3:6] / T[3:6, 3:6].sum(1, keepdims=True)
    w, v = np.linalg.eig(R[:3, :3].T)
    p = np.zeros(6)
    p[:3] = np.real(v[:, np.argmax(np.real(w))])
    check_transition_matrix(R, p / p.sum())
    check_transition_matrix(np.empty((0, 0)), np.empty((0,)))
