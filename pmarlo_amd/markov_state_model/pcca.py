"""PCCA+ metastable sets.

The reference wraps deeptime's ``pcca`` (pcca_like_macrostates / _canonicalize_macro_labels,
S/markov_state_model/_msm_utils.py:91-101, 284-299).  deeptime 0.4.5 is absent here (parity
unpinned); this module restates the published algorithm it implements:

* Deuflhard & Weber, Lin. Alg. Appl. 398 (2005): the m dominant right eigenvectors of a reversible T
  span (up to a linear map A) the membership functions; the inner simplex algorithm picks the m most
  outlying rows of the eigenvector matrix as vertices and takes A = (those rows)^-1;
* Roeblitz & Weber, Adv. Data Anal. Classif. 7 (2013): A is then refined by maximising the crispness
  sum_ij A_ij^2 / A_0j over the feasible set (memberships >= 0, rows sum to 1), with Nelder-Mead on
  the free (m-1) x (m-1) block as deeptime does (scipy.optimize.fmin).

The eigenvectors come from the device (msm_spectrum's left Ritz vectors; right = left / pi for a
reversible matrix); the simplex search and the optimisation act on k x m arrays on the host."""

from __future__ import annotations

import numpy as np

from ..device import get_engine

__all__ = ["pcca_memberships", "pcca_like_macrostates", "canonicalize_macro_labels"]


def _dominant_right_eigenvectors(T: np.ndarray, m: int, pi: np.ndarray | None):
    eng = get_engine()
    k = T.shape[0]
    spec = eng.spectrum(eng.to_device(T), n_its=0, n_vecs=m, tol=1e-10)
    pi_d = spec["pi"].to_host().ravel() if pi is None else np.asarray(pi, dtype=np.float64)
    ritz = spec["ritz"][0]
    if k > 1 and abs(ritz[1]) > 1.0 - 1e-12:
        raise ValueError("Transition matrix is disconnected (eigenvalue 1 is degenerate)")
    if np.any(pi_d <= 0):
        raise ValueError("Stationary distribution must be positive on every state")
    flux = pi_d[:, None] * T
    if not np.allclose(flux, flux.T, rtol=1e-5, atol=1e-15):
        raise ValueError("Transition matrix does not fulfill detailed balance")
    left = spec["vecs"].to_host()[0]                      # [m, k]
    if not np.all(np.isfinite(left)):
        raise ValueError("dominant eigenvalues are not real")
    R = (left / pi_d[None, :]).T                          # right eigenvectors of a reversible T
    R /= np.sqrt(np.einsum("i,ij,ij->j", pi_d, R, R))[None, :]
    R[:, 0] = np.abs(R[:, 0])                             # the constant vector, positive
    return R, pi_d


def _inner_simplex(R: np.ndarray) -> np.ndarray:
    """Rows of R that span the largest simplex: farthest row from the origin first, then m-1 rounds of
    'farthest from the span of the vertices so far' (Gram-Schmidt deflation).  Returns A = R[vertices]^-1."""
    k, m = R.shape
    work = R.copy()
    vertices = [int(np.argmax(np.einsum("ij,ij->i", work, work)))]
    work -= R[vertices[0]]
    for _ in range(1, m):
        pivot = work[vertices[-1]].copy()
        nrm = float(np.sqrt(pivot @ pivot))
        if nrm > 0.0:
            pivot /= nrm
            work -= np.outer(work @ pivot, pivot)
        dist = np.einsum("ij,ij->i", work, work)
        dist[vertices] = -1.0
        vertices.append(int(np.argmax(dist)))
    return np.linalg.inv(R[vertices])


def _complete(block: np.ndarray, R: np.ndarray) -> np.ndarray:
    """Feasible A from its free lower-right block: rows 1.. sum to zero (memberships sum to one), the top
    row is the smallest shift that keeps every membership non-negative, overall scale = partition of unity."""
    body = np.concatenate([-block.sum(axis=1, keepdims=True), block], axis=1)     # (m-1) x m
    top = np.max(-(R[:, 1:] @ body), axis=0, keepdims=True)                        # 1 x m
    A = np.concatenate([top, body], axis=0)
    return A / top.sum()


def _refine(R: np.ndarray, A0: np.ndarray) -> np.ndarray:
    from scipy.optimize import fmin

    m = A0.shape[0]
    if m < 2:
        return A0

    def neg_crispness(vec):
        A = _complete(vec.reshape(m - 1, m - 1), R)
        return -float(np.sum(A * A / A[0:1, :]))

    best = fmin(neg_crispness, A0[1:, 1:].ravel(), disp=False)
    return _complete(best.reshape(m - 1, m - 1), R)


def pcca_memberships(T: np.ndarray, m: int, pi: np.ndarray | None = None) -> np.ndarray:
    """Fuzzy memberships chi [k, m] (rows sum to 1, entries in [0, 1]) of the m metastable sets of a
    connected, reversible transition matrix.  ValueError when T is not such a matrix."""
    T = np.ascontiguousarray(T, dtype=np.float64)
    if T.ndim != 2 or T.shape[0] != T.shape[1]:
        raise ValueError("transition matrix must be square")
    k = T.shape[0]
    m = int(m)
    if m <= 0 or m > k:
        raise ValueError(f"Number of metastable sets must be in [1, {k}], got {m}")
    if np.any(T < 0) or not np.allclose(T.sum(axis=1), 1.0, atol=1e-10):
        raise ValueError("Input matrix is not a transition matrix")
    if m == 1:
        return np.ones((k, 1))
    R, _ = _dominant_right_eigenvectors(T, m, pi)
    A = _refine(R, _inner_simplex(R))
    chi = np.clip(R @ A, 0.0, 1.0)
    return chi / chi.sum(axis=1, keepdims=True)


def canonicalize_macro_labels(labels: np.ndarray, T: np.ndarray) -> np.ndarray:
    """Renumber macrostates by decreasing stationary population
    (_canonicalize_macro_labels, S/markov_state_model/_msm_utils.py:91-101)."""
    labels = np.asarray(labels)
    if labels.size == 0:
        return labels.astype(int)
    eng = get_engine()
    pi = eng.spectrum(eng.to_device(np.ascontiguousarray(T, dtype=np.float64)), n_its=0)["pi"].to_host().ravel()
    present = np.unique(labels)
    pops = np.asarray([pi[labels == u].sum() for u in present])
    rank = np.empty(present.size, dtype=int)
    rank[np.argsort(-pops, kind="stable")] = np.arange(present.size)
    return rank[np.searchsorted(present, labels)].astype(int)


def pcca_like_macrostates(T: np.ndarray, n_macrostates: int = 4, random_state: int | None = 42) -> np.ndarray | None:
    """Crisp macrostate label per microstate (argmax of the PCCA+ memberships, canonical numbering), or None
    when T is too small or PCCA+ rejects it (pcca_like_macrostates, S/markov_state_model/_msm_utils.py:284-299).
    `random_state` is accepted for signature compatibility; nothing here is stochastic."""
    T = np.asarray(T, dtype=np.float64)
    if T.size == 0 or T.shape[0] <= int(n_macrostates):
        return None
    try:
        chi = pcca_memberships(T, int(n_macrostates))
    except ValueError:
        return None
    return canonicalize_macro_labels(np.argmax(chi, axis=1).astype(int), T)
