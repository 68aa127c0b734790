"""HIP transition matrix / stationary distribution / spectrum / implied timescales vs the
oracle (numpy restatement of the reference's estimators; tolerances per north_star: 1e-5
relative on timescales, held much tighter here)."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import cport, npport
from tests import _gen

pytestmark = pytest.mark.gpu


def _metastable_counts(k, n, lag, seed, n_macro=5):
    lab = _gen.metastable_labels(n, k, n_macro, seed=seed)
    c, _ = cport.count_transitions(lab, k, lag)
    return c, lab


def test_normalise_counts_mode0(engine, golden):
    g = golden("counts.npz")
    C = g["plain_counts"].astype(np.int64)
    out = engine.transition_matrix(engine.to_device(C), mode=0)
    np.testing.assert_array_equal(out["T"].to_host(), g["normalised_plain"])
    # zero rows stay zero; diag mass = trace/k
    C2 = C.copy()
    C2[3, :] = 0
    out2 = engine.transition_matrix(engine.to_device(C2), mode=0)
    T2 = out2["T"].to_host()
    np.testing.assert_array_equal(T2, npport.normalise_counts(C2))
    assert np.all(T2[3] == 0)
    np.testing.assert_allclose(out2["diag_mass"].to_host()[0], np.trace(T2) / 7, rtol=1e-15)
    # float64 (weighted) counts
    Cw = g["weighted_counts"]
    # non-integer row sums depend on the summation order at the last bit
    np.testing.assert_allclose(engine.transition_matrix(engine.to_device(Cw), mode=0)["T"].to_host(),
                               npport.normalise_counts(Cw), rtol=1e-15)


def test_connected_mode1_matches_ml_msm(engine):
    k = 40
    C, _ = _metastable_counts(k, 30_000, 5, seed=3, n_macro=4)
    C[:, 7] = 0
    C[7, :] = 0          # disconnected state -> inactive
    C[:, 21] = 0
    C[21, :] = 0
    out = engine.transition_matrix(engine.to_device(C), mode=1)
    ka = int(out["n_active"].to_host()[0])
    ref = npport.ml_msm(C)
    assert ka == ref["active"].size == k - 2
    np.testing.assert_array_equal(out["active"].to_host()[:ka], ref["active"])
    Ta = out["T"].to_host()[:ka, :ka]
    Ca, _ = npport.ensure_connected_counts(C)
    np.testing.assert_allclose(Ta, Ca / Ca.sum(axis=1, keepdims=True), rtol=1e-14)
    spec = engine.spectrum(out["T"], n=out["n_active"], n_its=3, lags=[5.0])
    pi = spec["pi"].to_host()[0, :ka]
    np.testing.assert_allclose(pi, ref["stationary_distribution"][ref["active"]], rtol=1e-8, atol=1e-12)
    T_full, pi_full = engine.embed_full(out["T"], out["inv_map"], spec["pi"])
    np.testing.assert_allclose(T_full.to_host(), ref["transition_matrix"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(pi_full.to_host(), ref["stationary_distribution"], rtol=1e-8, atol=1e-12)
    assert pi_full.to_host()[7] == 0 and T_full.to_host()[7, 7] == 1.0


def test_two_state_chain_analytic(engine):
    """tests/unit/markov_state_model/test_two_state_msm.py:6-22: t2 = -1/ln(0.8)."""
    T = np.array([[0.9, 0.1], [0.1, 0.9]])
    spec = engine.spectrum(engine.to_device(T), n_its=1, lags=[1.0])
    np.testing.assert_allclose(spec["its_eig"][0], [0.8], rtol=1e-12)
    np.testing.assert_allclose(spec["its_ts"][0], [-1.0 / np.log(0.8)], rtol=1e-12)
    np.testing.assert_allclose(spec["pi"].to_host()[0], [0.5, 0.5], rtol=1e-12)


@pytest.mark.parametrize("k,n,lag,n_its", [(20, 50_000, 1, 3), (100, 200_000, 10, 5), (200, 300_000, 5, 4),
                                            (500, 1_000_000, 10, 6)])
def test_its_vs_oracle(engine, k, n, lag, n_its):
    C, _ = _metastable_counts(k, n, lag, seed=k)
    out = engine.transition_matrix(engine.to_device(C), mode=1)
    spec = engine.spectrum(out["T"], n=out["n_active"], n_its=n_its, lags=[float(lag)])
    ev_ref, ts_ref = npport.its_from_counts(C, lag, n_its)
    # north_star: implied timescales within 1e-5 relative
    np.testing.assert_allclose(spec["its_eig"][0], ev_ref, rtol=1e-8)
    np.testing.assert_allclose(spec["its_ts"][0], ts_ref, rtol=1e-6)
    ka = int(out["n_active"].to_host()[0])
    Ta = out["T"].to_host()[:ka, :ka]
    pi = spec["pi"].to_host()[0, :ka]
    assert pi.min() > 0
    np.testing.assert_allclose(pi @ Ta, pi, atol=1e-12)
    np.testing.assert_allclose(pi.sum(), 1.0, rtol=1e-13)


def test_spectrum_matches_numpy_eigvals_including_complex(engine):
    """A non-reversible (cyclic-drift) chain has complex leading eigenvalues."""
    k = 30
    rng = np.random.default_rng(5)
    T = np.full((k, k), 1e-3)
    for i in range(k):
        T[i, i] += 0.5
        T[i, (i + 1) % k] += 0.45     # strong drift
        T[i, (i - 1) % k] += 0.02
    T += 0.01 * rng.random((k, k))
    T /= T.sum(axis=1, keepdims=True)
    spec = engine.spectrum(engine.to_device(T), p=12, n_its=0, tol=1e-11)
    ref = np.linalg.eigvals(T)
    ref = ref[np.argsort(-np.abs(ref))]
    got = spec["ritz"][0][:5]
    assert np.abs(ref[1].imag) > 1e-3
    # complex pairs are monitored by value change, not residual: hold them to the
    # north_star tolerance (1e-5 relative) with a decade to spare
    np.testing.assert_allclose(np.abs(got), np.abs(ref[:5]), rtol=1e-6)
    np.testing.assert_allclose(np.sort(got[:5].real), np.sort(ref[:5].real), atol=1e-6)


def test_lagscan_batch(engine):
    """C4-like: k=200, 20 lags in one batched solve (ITS scan, _its.py:272-357)."""
    k, n = 200, 200_000
    lab = _gen.metastable_labels(n, k, 5, seed=9)
    lags = [1, 2, 3, 5, 8, 10, 15, 20, 30, 50]
    ld = engine.to_device(lab)
    counts, _ = engine.count_transitions_lagscan(ld, k, lags)
    Ts, ns = [], []
    Tb = engine.empty((len(lags), k, k), np.float64)
    nb = engine.empty((len(lags),), np.int32)
    for i in range(len(lags)):
        out = engine.transition_matrix(counts.view((k, k), offset_elems=i * k * k), mode=1)
        Tb.view((k, k), offset_elems=i * k * k).copy_from_host(out["T"].to_host())
        nb.view((1,), offset_elems=i).copy_from_host(out["n_active"].to_host())
    spec = engine.spectrum(Tb, n=nb, n_its=3, lags=[float(v) for v in lags])
    ch = counts.to_host()
    for i, lag in enumerate(lags):
        ev_ref, ts_ref = npport.its_from_counts(ch[i], lag, 3)
        np.testing.assert_allclose(spec["its_ts"][i], ts_ref, rtol=1e-6)


def test_clustered_spectrum_is_reported_not_guessed(engine):
    """Diffusion on a ring packs dozens of eigenvalues within 1e-3 of 1: subspace iteration
    cannot resolve them in its budget and must say so instead of returning stale Ritz values."""
    from pmarlo_amd._lib import MsmError

    k = 300
    lab = _gen.markov_labels(400_000, k, seed=1, stay=0.97)
    C, _ = cport.count_transitions(lab, k, 1)
    out = engine.transition_matrix(engine.to_device(C), mode=1)
    with pytest.raises(MsmError, match="residual"):
        engine.spectrum(out["T"], n=out["n_active"], n_its=4, lags=[1.0], max_launches=3)
    loose = engine.spectrum(out["T"], n=out["n_active"], n_its=4, lags=[1.0], max_launches=3, allow_unconverged=True)
    assert loose["residual"][0] > 1e-9


@pytest.mark.parametrize("k,orders,squarings", [(70, [70, 33, 1, 64], 1), (200, [200, 150, 199], 2), (500, [500], 2),
                                                (129, [128, 129], 3)])
def test_matrix_power_matches_numpy(engine, k, orders, squarings):
    """msm_matrix_power: T^(2^s) of a ragged batch on the fp64 matrix cores; zeros outside each matrix's block."""
    rng = np.random.default_rng(k)
    B = len(orders)
    T = np.zeros((B, k, k))
    for b, n in enumerate(orders):
        A = rng.random((n, n)) ** 3
        T[b, :n, :n] = A / A.sum(1, keepdims=True)
    nd = engine.to_device(np.asarray(orders, np.int32))
    got = engine.matrix_power(engine.to_device(T), squarings, n=nd).to_host()
    want = np.stack([np.linalg.matrix_power(T[b], 2 ** squarings) for b in range(B)])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-14)
    for b, n in enumerate(orders):
        assert not got[b, n:, :].any() and not got[b, :, n:].any()
    single = engine.matrix_power(engine.to_device(T[0]), squarings).to_host()
    np.testing.assert_allclose(single, want[0], rtol=0, atol=1e-14)


def test_spectrum_on_a_power_reports_the_values_of_T(engine):
    """The iterations may run on T^4; Ritz values, timescales and pi are those of T (against numpy, and against the run on T)."""
    rng = np.random.default_rng(3)
    k = 160
    P = np.full((k, k), 1e-3)
    w = k // 4
    for b in range(4):
        P[w * b:w * b + w, w * b:w * b + w] += rng.random((w, w)) ** 2
    P /= P.sum(1, keepdims=True)
    Td = engine.to_device(P)
    a = engine.spectrum(Td, n_its=4, lags=[5.0], squarings=2)
    b_ = engine.spectrum(Td, n_its=4, lags=[5.0], squarings=0)
    ev = np.linalg.eigvals(P)
    ev = ev[np.argsort(-np.abs(ev))][:5].real
    np.testing.assert_allclose(np.sort(a["its_eig"][0])[::-1], np.sort(np.abs(ev[1:5]))[::-1], rtol=1e-9)   # |real part|, as msm_spectrum reports it
    np.testing.assert_allclose(a["its_eig"], b_["its_eig"], rtol=1e-9)
    np.testing.assert_allclose(a["its_ts"], b_["its_ts"], rtol=1e-9)
    np.testing.assert_allclose(a["pi"].to_host(), b_["pi"].to_host(), rtol=1e-9, atol=1e-14)
    assert a["launches"] <= b_["launches"]


@pytest.mark.parametrize("k,rank", [(40, 3), (64, 6), (120, 2)])
def test_spectrum_of_low_rank_chains_with_and_without_powering(engine, k, rank):
    """Matrices whose spectrum below the watched values is (numerically) zero: T^4 squashes those directions to the
    rounding level, the Cholesky-QR of such a basis is the delicate case of the powered iteration.  The reported values
    must equal numpy's either way (the powered attempt hands over to the plain iteration when it cannot converge)."""
    rng = np.random.default_rng(k + rank)
    memb = rng.random((k, rank)) ** 2
    A = memb @ memb.T + 1e-3 * np.eye(k)                     # symmetric: a reversible chain, real spectrum
    P = A / A.sum(1, keepdims=True)                          # `rank` eigenvalues of order 1, the rest ~1e-3 / row sum
    n_its = min(4, rank - 1)
    ev = np.sort(np.abs(np.linalg.eigvals(P).real))[::-1][1:1 + n_its]
    for sq in (0, 2):
        out = engine.spectrum(engine.to_device(P), n_its=n_its, lags=[1.0], squarings=sq)
        np.testing.assert_allclose(np.sort(out["its_eig"][0])[::-1], ev, rtol=1e-7)


def test_complex_ritz_pairs_get_a_true_residual(engine):
    """A three-block chain with a cyclic drift between the blocks: the slowest processes are a COMPLEX pair.  Its
    convergence is judged by the residual of the real invariant plane (not by the change between two launches), so
    the solve may finish in its first launch, and the reported residual must be honest: the Ritz values agree with
    numpy to the tolerance the residual promises."""
    rng = np.random.default_rng(5)
    k, w = 96, 32
    P = np.zeros((k, k))
    for b in range(3):
        blk = rng.random((w, w)) + 0.2
        P[w * b:w * b + w, w * b:w * b + w] = blk / blk.sum(1, keepdims=True)
    eps = 0.02
    shift = np.roll(np.eye(k), w, axis=1)                      # state i of block b -> state i of block b + 1
    T = (1.0 - eps) * P + eps * shift
    ev = np.linalg.eigvals(T)
    ev = ev[np.argsort(-np.abs(ev))]
    assert abs(ev[1].imag) > 1e-3 and abs(ev[1] - np.conj(ev[2])) < 1e-12        # the pair the test is about
    for sq in (0, 2):
        out = engine.spectrum(engine.to_device(T), n_its=2, lags=[1.0], squarings=sq, tol=1e-10)
        got = out["ritz"][0][:3]
        assert abs(got[0] - 1.0) < 1e-9
        assert min(abs(got[1] - ev[1]), abs(got[1] - ev[2])) < 1e-8
        assert abs(got[1] - np.conj(got[2])) < 1e-12
        assert float(out["residual"][0]) <= 1e-10
        assert out["launches"] <= 3


@pytest.mark.parametrize("k", [4, 7, 12])
def test_small_singular_matrices_need_no_iterations(engine, k):
    """k below the subspace size: the basis spans the whole space, the Rayleigh-Ritz step alone gives the spectrum, and a
    singular T (two states with identical rows) must not meet the Cholesky-QR of T'Z, whose Gram matrix it makes singular."""
    rng = np.random.default_rng(k)
    P = rng.random((k, k)) + 0.05
    P[1] = P[0]                                   # rank k - 1: an exact zero eigenvalue
    P /= P.sum(1, keepdims=True)
    n_its = k - 1
    out = engine.spectrum(engine.to_device(P), n_its=n_its, lags=[1.0])
    assert out["launches"] == 1
    ev = np.linalg.eigvals(P)
    want = np.sort(np.abs(np.sort(ev.real)[::-1][1:1 + n_its]))[::-1]
    got = np.sort(np.nan_to_num(out["its_eig"][0], nan=0.0))[::-1]
    real = np.abs(ev.imag) < 1e-12
    if real.all():
        np.testing.assert_allclose(got, np.clip(want, 1e-12, None), atol=1e-9)
    np.testing.assert_allclose(out["pi"].to_host()[0] @ P, out["pi"].to_host()[0], atol=1e-12)
