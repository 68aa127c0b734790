"""PCCA+ on the engine's eigenvectors (msm_spectrum d_vecs) against the numpy restatement of the
published algorithm (oracle/npport.pcca_memberships; deeptime's pcca is absent: parity unpinned)
and against what defines metastable sets: block recovery, partition of unity, canonical numbering."""
import numpy as np
import pytest

from oracle import npport
from pmarlo_amd.markov_state_model import fit_reversible_msm
from pmarlo_amd.markov_state_model.pcca import canonicalize_macro_labels, pcca_like_macrostates, pcca_memberships

pytestmark = pytest.mark.gpu


def _block_counts(sizes, seed, leak=0.02):
    rng = np.random.default_rng(seed)
    k = sum(sizes)
    C = rng.random((k, k)) * leak
    o = 0
    for s in sizes:
        C[o:o + s, o:o + s] += rng.random((s, s)) + 0.3
        o += s
    return C


def _reversible_T(sizes, seed):
    C = _block_counts(sizes, seed)
    C = C + C.T
    return C / C.sum(1, keepdims=True)


def test_left_ritz_vectors(engine):
    T = _reversible_T([12, 9, 7, 5], 1)
    spec = engine.spectrum(engine.to_device(T), n_its=0, n_vecs=5, tol=1e-11)
    got = spec["vecs"].to_host()[0]
    w, V = np.linalg.eig(T.T)
    order = np.argsort(-np.abs(w))[:5]
    for q, idx in enumerate(order):
        ref = np.real(V[:, idx])
        ref /= np.linalg.norm(ref)
        if ref[np.argmax(np.abs(ref))] < 0:
            ref = -ref
        np.testing.assert_allclose(got[q], ref, atol=1e-8)
    np.testing.assert_allclose(np.real(spec["ritz"][0][:5]), np.real(w[order]), rtol=1e-9)
    # a rotation-like (non-reversible) matrix has complex pairs: NaN rows, not garbage
    P = np.roll(np.eye(6), 1, axis=1) * 0.9 + 0.1 / 6
    out = engine.spectrum(engine.to_device(P), n_its=0, n_vecs=3, allow_unconverged=True)
    v, ritz = out["vecs"].to_host()[0], out["ritz"][0]
    assert np.all(np.isfinite(v[0]))
    # -0.9 and the two complex pairs share the modulus 0.9: their order among the Ritz values is a matter of the last bit
    assert np.iscomplex(ritz[1:3]).any()
    for q in (1, 2):
        assert np.all(np.isnan(v[q])) if ritz[q].imag != 0.0 else np.all(np.isfinite(v[q]))


@pytest.mark.parametrize("sizes", [[10, 10], [12, 9, 7, 5], [40, 30, 20, 10, 6, 4]])
def test_pcca_recovers_blocks_and_matches_oracle(engine, sizes):
    T = _reversible_T(sizes, len(sizes))
    m = len(sizes)
    chi = pcca_memberships(T, m)
    assert chi.shape == (T.shape[0], m)
    np.testing.assert_allclose(chi.sum(1), 1.0, rtol=1e-12)
    assert chi.min() >= 0.0 and chi.max() <= 1.0
    want = npport.pcca_memberships(T, m)
    np.testing.assert_allclose(chi, want, atol=2e-4)        # Nelder-Mead from starts that differ by 1e-10
    truth = np.repeat(np.arange(m), sizes)
    lab = np.argmax(chi, axis=1)
    for b in range(m):                                       # every block is one set
        assert np.unique(lab[truth == b]).size == 1
    assert np.unique(lab).size == m
    assert chi.max(axis=1).min() > 0.8                       # crisp for a well-separated chain
    labels = pcca_like_macrostates(T, n_macrostates=m)
    np.testing.assert_array_equal(labels, canonicalize_macro_labels(lab, T))
    pops = np.asarray([npport.stationary_distribution(T)[labels == q].sum() for q in range(m)])
    assert np.all(np.diff(pops) <= 1e-12)                    # numbered by decreasing population


def test_pcca_on_an_estimated_reversible_msm(engine):
    C = np.round(_block_counts([15, 10, 8], 7, leak=0.05) * 50)
    T, pi, active = fit_reversible_msm(C)
    labels = pcca_like_macrostates(T, n_macrostates=3)
    assert labels is not None and np.unique(labels).size == 3
    truth = np.repeat(np.arange(3), [15, 10, 8])
    for b in range(3):
        assert np.unique(labels[truth == b]).size == 1


def test_pcca_rejections(engine):
    C = _block_counts([8, 8], 3)
    T_nonrev = C / C.sum(1, keepdims=True)
    assert pcca_like_macrostates(T_nonrev, n_macrostates=2) is None          # no detailed balance
    with pytest.raises(ValueError, match="detailed balance"):
        pcca_memberships(T_nonrev, 2)
    T = _reversible_T([5, 5], 2)
    assert pcca_like_macrostates(T[:3, :3] / T[:3, :3].sum(1, keepdims=True), n_macrostates=4) is None   # too small
    assert pcca_like_macrostates(np.empty((0, 0))) is None
    with pytest.raises(ValueError):
        pcca_memberships(T, 0)
    with pytest.raises(ValueError, match="not a transition matrix"):
        pcca_memberships(T * 1.1, 2)
    D = np.zeros((10, 10))
    D[:5, :5] = _reversible_T([5], 1)
    D[5:, 5:] = _reversible_T([5], 2)
    with pytest.raises(ValueError, match="disconnected"):
        pcca_memberships(D, 2)
    np.testing.assert_array_equal(pcca_memberships(T, 1), np.ones((10, 1)))
