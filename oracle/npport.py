"""numpy / scipy / scikit-learn restatement of the floating-point operators on
pmarlo's featurize -> TICA -> k-means -> T-matrix -> ITS path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Paths cited as S/... are
relative to the reference tree (S/ = src/pmarlo/).

Pinning status (tests/test_oracle_golden.py, vectors from tests/golden/make_golden.py
which imports the reference in the build container):
  pinned by reference import : weighted_counts, expected_pairs, state counts,
      normalise_counts, preprocess, estimate_top_eigenvalues, kmeans discretizer
      (fit + transform, both sklearn branches), discretize_dataset, safe_timescales,
      featurizer geometry (distance / angle / dihedral).
  PARITY UNPINNED (third-party code absent from the container, restated from the
  published algorithm of the pinned version): deeptime 0.4.5 TICA (tica_fit /
  tica_reduce), deeptime MaximumLikelihoodMSM(reversible=False) + stationary
  distribution + eigenvalues (ml_msm / its_from_counts), mdtraj geometry (replaced
  by the in-repo torch extractor formula, which IS pinned).
"""

from __future__ import annotations

from typing import Iterable, Sequence

import numpy as np

NUMERIC_MIN_POSITIVE = 1e-12       # S/constants.py:29
NUMERIC_DIRICHLET_ALPHA = 1e-3     # S/constants.py:44


# ---------------------------------------------------------------------------
# counts (S/analysis/discretize.py:596-682, S/analysis/counting.py:10-68)
# ---------------------------------------------------------------------------
def weighted_counts(labels, n_states, lag_time, weights=None, segments=None, stride=1):
    """_weighted_counts, S/analysis/discretize.py:609-645."""
    labels = np.asarray(labels)
    counts = np.zeros((n_states, n_states), dtype=np.float64)
    if labels.size == 0 or lag_time <= 0:
        return counts, 0
    total = 0
    step = max(1, int(stride))
    segs = [(0, labels.size)] if segments is None else [
        (max(0, int(a)), min(labels.size, int(b))) for a, b in segments]
    for start, stop in segs:
        if stop <= start or stop - start <= lag_time:
            continue
        src = labels[start:stop - lag_time:step]
        dst = labels[start + lag_time:stop:step]
        if src.size == 0:
            continue
        valid = (src >= 0) & (dst >= 0)
        if not np.any(valid):
            continue
        if weights is not None:
            np.add.at(counts, (src[valid], dst[valid]), np.asarray(weights)[start:stop - lag_time:step][valid])
        else:
            np.add.at(counts, (src[valid], dst[valid]), 1.0)
        total += int(np.count_nonzero(valid))
    return counts, total


def expected_pairs(lengths: Iterable[int], tau: int, stride=1) -> int:
    """S/analysis/counting.py:10-68."""
    lengths = [int(v) for v in lengths]
    strides = [int(v) for v in stride] if isinstance(stride, Iterable) else [int(stride)]
    total = 0
    for idx, length in enumerate(lengths):
        eff = length - tau
        if length <= 0 or eff <= 0:
            continue
        step = strides[idx] if idx < len(strides) else strides[-1]
        total += 1 + (eff - 1) // step
    return total


def normalise_counts(C):
    """_normalise_counts, S/analysis/discretize.py:678-682 (zero rows stay 0)."""
    C = np.asarray(C, dtype=np.float64)
    rs = C.sum(axis=1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.divide(C, rs, out=np.zeros_like(C), where=rs > 0)


# ---------------------------------------------------------------------------
# preprocessing + TICA
# ---------------------------------------------------------------------------
def preprocess(X, scale=True):
    """reduction._preprocess, S/markov_state_model/reduction.py:13-40.

    SimpleImputer(mean) (all-NaN column -> 0) then StandardScaler(with_mean=True,
    with_std=scale): population std (ddof=0), zero std -> 1."""
    Xp = np.array(X, dtype=float)
    if Xp.size == 0:
        return np.zeros_like(Xp)
    nan = np.isnan(Xp)
    if nan.any():
        cnt = (~nan).sum(axis=0)
        col_mean = np.where(cnt > 0, np.where(nan, 0.0, Xp).sum(axis=0) / np.maximum(cnt, 1), 0.0)
        Xp = np.where(nan, col_mean[None, :], Xp)
    mean = Xp.mean(axis=0)
    out = Xp - mean
    if scale:
        std = np.sqrt(((Xp - mean) ** 2).mean(axis=0))
        std = np.where(std < 10 * np.finfo(float).eps, 1.0, std)
        out = out / std
    return np.nan_to_num(out, nan=0.0)


def estimate_top_eigenvalues(outputs, idx_t, idx_tau, n_out):
    """_estimate_top_eigenvalues, S/features/deeptica/core/trainer_api.py:632-656."""
    y_t = outputs[idx_t]
    y_tau = outputs[idx_tau]
    y_t_c = y_t - y_t.mean(axis=0, keepdims=True)
    y_tau_c = y_tau - y_tau.mean(axis=0, keepdims=True)
    n = max(1, y_t_c.shape[0] - 1)
    C0 = (y_t_c.T @ y_t_c) / float(n)
    Ct = (y_t_c.T @ y_tau_c) / float(n)
    evals, evecs = np.linalg.eigh((C0 + C0.T) * 0.5)
    evals = np.clip(evals, NUMERIC_MIN_POSITIVE, None)
    inv_sqrt = evecs @ np.diag(1.0 / np.sqrt(evals)) @ evecs.T
    M = inv_sqrt @ Ct @ inv_sqrt.T
    eigs = np.sort(np.linalg.eigvalsh((M + M.T) * 0.5))[::-1]
    return eigs[: min(int(n_out), eigs.size)]


def _sort_desc_by_norm(evals, evecs):
    order = np.argsort(np.abs(evals))[::-1]
    return evals[order], evecs[:, order]


def lagged_moments(Xs: Sequence[np.ndarray], lag: int):
    """Raw symmetric moments of deeptime's Covariance(lagtime, remove_data_mean=True,
    reversible=True, bessels_correction=False) -- the estimator TICA.fit uses
    (deeptime 0.4.5 decomposition/_tica.py, covariance/util/_running_moments.py).
    x = X[:-lag], y = X[lag:] per trajectory; w = 2T;
    mean = (sum x + sum y)/w; C00 = (xc'xc + yc'yc)/w; C0t = (xc'yc + yc'xc)/w."""
    F = Xs[0].shape[1]
    T = 0
    sx = np.zeros(F)
    sy = np.zeros(F)
    Mxx = np.zeros((F, F))
    Mxy = np.zeros((F, F))
    for X in Xs:
        X = np.asarray(X, dtype=np.float64)
        if X.shape[0] <= lag:
            continue
        x, y = X[:-lag], X[lag:]
        T += x.shape[0]
        sx += x.sum(axis=0)
        sy += y.sum(axis=0)
        Mxx += x.T @ x + y.T @ y
        Mxy += x.T @ y
    return {"T": T, "sx": sx, "sy": sy, "Mxx": Mxx, "Mxy_half": Mxy}


def tica_fit(Xs: Sequence[np.ndarray], lag: int, dim: int | None = None, epsilon: float = 1e-6,
             scaling: str | None = "kinetic_map"):
    """deeptime.decomposition.TICA(lagtime, dim).fit(list).fetch_model() restated
    (call sites S/markov_state_model/reduction.py:103-109, _features.py:200-202).

    Steps (deeptime 0.4.5): reversible covariances; spd_inv_split(C00, eps) via eigh,
    eigenvalues sorted by descending magnitude, cut-off eps (raised to |min ev| when
    C00 has negative eigenvalues), canonical signs (largest |component| positive);
    eigh of L' C0t L; sort descending by magnitude; eigenvectors R = L R', canonical
    signs; kinetic_map scaling multiplies eigenvector i by eigenvalue i.  PARITY UNPINNED."""
    return tica_from_moments(lagged_moments(Xs, lag), dim=dim, epsilon=epsilon, scaling=scaling)


def tica_from_moments(m, dim=None, epsilon: float = 1e-6, scaling: str | None = "kinetic_map", scale=None):
    """The solve of tica_fit from raw moments (the form shards exchange).  `scale` (per-feature divisor):
    covariances of the data divided by it, i.e. of the standardised features."""
    T = m["T"]
    if T == 0:
        raise ValueError("no lagged pairs")
    w = 2.0 * T
    isc = np.ones_like(m["sx"]) if scale is None else 1.0 / np.asarray(scale, np.float64)
    ss = np.outer(isc, isc)
    mean = (m["sx"] + m["sy"]) / w * isc
    C00 = m["Mxx"] / w * ss - np.outer(mean, mean)
    Mxy = m["Mxy_half"] + m["Mxy_half"].T
    C0t = Mxy / w * ss - np.outer(mean, mean)
    C00 = 0.5 * (C00 + C00.T)
    C0t = 0.5 * (C0t + C0t.T)
    s, V = np.linalg.eigh(C00)
    s, V = _sort_desc_by_norm(s, V)
    eps = epsilon
    if s.min() < 0:
        eps = max(eps, -s.min() + 1e-16)
    rank = int(s.shape[0] - np.searchsorted(np.abs(s)[::-1], eps))
    if rank == 0:
        raise ValueError("C00 has zero rank")
    Vm, sm = V[:, :rank].copy(), s[:rank]
    for j in range(rank):
        Vm[:, j] *= np.sign(Vm[np.argmax(np.abs(Vm[:, j])), j])
    L = Vm / np.sqrt(sm)[None, :]
    Ct = L.T @ C0t @ L
    lam, Rt = np.linalg.eigh(0.5 * (Ct + Ct.T))
    lam, Rt = _sort_desc_by_norm(lam, Rt)
    R = L @ Rt
    for j in range(R.shape[1]):
        R[:, j] *= np.sign(R[np.argmax(np.abs(R[:, j])), j])
    if scaling in ("kinetic_map", "km"):
        R = R * lam[None, :]
    out_dim = rank if dim is None else min(int(dim), rank)
    return {"mean": mean, "eigenvalues": lam, "coefficients": R, "dim": out_dim, "rank": rank,
            "C00": C00, "C0t": C0t, "T": T}


def tica_transform(model, X):
    return (np.asarray(X, dtype=np.float64) - model["mean"]) @ model["coefficients"][:, : model["dim"]]


def tica_reduce(X, lag=1, n_components=2, scale=True):
    """reduction.tica_reduce, S/markov_state_model/reduction.py:77-110."""
    Xp = preprocess(X, scale=scale)
    model = tica_fit([Xp], lag, dim=n_components)
    return tica_transform(model, Xp)


# ---------------------------------------------------------------------------
# k-means (sklearn branch of the reference, S/analysis/discretize.py:406-514)
# ---------------------------------------------------------------------------
def kmeans_discretizer_fit(X, n_states, random_state=None):
    """_KMeansDiscretizer.fit: mean, std(ddof=1), sklearn (MiniBatch)KMeans on whitened X."""
    from sklearn.cluster import KMeans, MiniBatchKMeans

    X = np.asarray(X, dtype=np.float64)
    mean = X.mean(axis=0)
    std = X.std(axis=0, ddof=1)
    std_safe = np.where(std > 1e-10, std, 1.0)
    Xz = (X - mean) / std_safe
    if X.shape[0] * X.shape[1] >= 5_000_000:
        model = MiniBatchKMeans(n_clusters=int(n_states), random_state=random_state)
    else:
        model = KMeans(n_clusters=int(n_states), random_state=random_state, n_init=10)
    model.fit(Xz)
    return {"mean": mean, "std": std, "std_safe": std_safe,
            "centers": np.asarray(model.cluster_centers_, dtype=np.float64), "model": model}


def kmeans_predict(Xz, centers):
    """argmin_j(|c_j|^2 - 2 x.c_j): sklearn predict's formula (BLAS summation order)."""
    csq = np.einsum("ij,ij->i", centers, centers)
    d = csq[None, :] - 2.0 * (Xz @ centers.T)
    return np.argmin(d, axis=1).astype(np.int32)


# ---------------------------------------------------------------------------
# MSM estimation + ITS
# ---------------------------------------------------------------------------
def ensure_connected_counts(C, alpha=NUMERIC_DIRICHLET_ALPHA, epsilon=NUMERIC_MIN_POSITIVE):
    """S/utils/msm_utils.py:129-167: active = rowsum+colsum > eps; +alpha on every active cell."""
    C = np.asarray(C, dtype=float)
    totals = C.sum(axis=1) + C.sum(axis=0)
    active = np.where(totals > epsilon)[0]
    if active.size == 0:
        return np.empty((0, 0)), active
    return C[np.ix_(active, active)] + float(alpha), active


def stationary_distribution(T):
    """Left eigenvector of T for eigenvalue 1, normalised to sum 1 (what deeptime's
    stationary_distribution returns for an irreducible dense matrix).  PARITY UNPINNED."""
    import scipy.linalg

    w, vl = scipy.linalg.eig(T, left=True, right=False)
    i = int(np.argmin(np.abs(w - 1.0)))
    pi = np.real(vl[:, i])
    return pi / pi.sum()


def ml_msm(counts, n_states=None):
    """_finalize_transition_and_stationary, S/markov_state_model/_estimation.py:158-188:
    MaximumLikelihoodMSM(reversible=False) on the regularised active block = row
    normalisation; T_full = I outside the active block, pi_full = 0 there."""
    C = np.asarray(counts, dtype=float)
    n = C.shape[0] if n_states is None else int(n_states)
    Ca, active = ensure_connected_counts(C)
    T_full = np.eye(n)
    pi_full = np.zeros(n)
    cm = np.zeros((n, n))
    if Ca.size:
        T = Ca / Ca.sum(axis=1, keepdims=True)
        pi = stationary_distribution(T)
        T_full[np.ix_(active, active)] = T
        pi_full[active] = pi
        cm[np.ix_(active, active)] = Ca
    return {"count_matrix": cm, "transition_matrix": T_full, "stationary_distribution": pi_full,
            "active": active}


def safe_timescales(lag, eigvals, eps=NUMERIC_MIN_POSITIVE):
    """S/markov_state_model/utils.py:17-57."""
    eig = np.asarray(eigvals)
    if eig.size == 0:
        return np.empty_like(eig, dtype=np.float64)
    ec = eig.astype(np.complex128)
    mag = np.abs(ec)
    with np.errstate(divide="ignore", invalid="ignore"):
        ts = -float(lag) / np.log(np.clip(mag, eps, 1 - eps))
    invalid = ~np.isfinite(mag) | (mag <= 0) | (mag >= 1)
    real = np.isclose(ec.imag, 0.0)
    invalid |= real & ((ec.real <= 0.0) | (ec.real >= 1.0))
    ts = np.asarray(ts, dtype=np.float64)
    ts[invalid] = np.nan
    return ts


def its_from_transition_matrix(T, lag, n_timescales):
    """_summarize_its_stats applied to ONE transition matrix
    (S/markov_state_model/_its.py:543-604): top (n+1) eigenvalues by magnitude
    (deeptime eigenvalues(T, k)), re-sorted by descending real part, drop the first,
    real part -> abs -> clip to [1e-12, 1-1e-12], safe_timescales."""
    T = np.asarray(T, dtype=float)
    n_eval = int(max(0, n_timescales))
    ev = np.linalg.eigvals(T)
    ev = ev[np.argsort(np.abs(ev))[::-1]]
    k_req = n_eval + 1 if n_eval > 0 else 1
    if k_req < T.shape[0]:
        ev = ev[:k_req]
    ev = ev[np.argsort(-np.real(ev))]
    eig = np.real(ev[1:1 + n_eval])
    eig = np.clip(np.abs(eig), NUMERIC_MIN_POSITIVE, 1.0 - NUMERIC_MIN_POSITIVE)
    ts = safe_timescales(int(max(1, lag)), eig)
    out_e = np.full(n_eval, np.nan)
    out_t = np.full(n_eval, np.nan)
    out_e[: eig.size] = eig
    out_t[: ts.size] = ts
    return out_e, out_t


def its_from_counts(C, lag, n_timescales, alpha=NUMERIC_DIRICHLET_ALPHA):
    """Deterministic ITS of SURVEY.md hard part 4: T = rownorm(C_active + alpha)."""
    Ca, _ = ensure_connected_counts(C, alpha=alpha)
    T = Ca / Ca.sum(axis=1, keepdims=True)
    return its_from_transition_matrix(T, lag, n_timescales)


def reversible_its_from_counts(C, lag, n_timescales):
    """Mathematically intended form of _deterministic_its_from_counts
    (S/markov_state_model/_its.py:742-801): C_rev = (C+C')/2, T = rownorm(C_rev),
    eigenvalues through the pi-symmetrised matrix.  The reference takes pi from the row
    sums of T (identically 1 -> uniform), which de-symmetrises the similarity
    transform; that quirk is NOT replicated."""
    C = np.asarray(C, dtype=float)
    Cr = 0.5 * (C + C.T)
    row = Cr.sum(axis=1)
    rs = np.where(row == 0, 1.0, row)
    dinv = 1.0 / np.sqrt(rs)
    S = Cr * dinv[:, None] * dinv[None, :]
    ev = np.linalg.eigvalsh(S)
    ev = ev[np.argsort(-np.abs(ev))]                        # deeptime eigenvalues(T, k): the k largest in magnitude
    if n_timescales + 1 < ev.size:
        ev = ev[:n_timescales + 1]
    ev = np.sort(ev)[::-1]                                  # then by descending value, first (= 1) dropped
    eig = np.clip(np.abs(ev[1:1 + n_timescales]), NUMERIC_MIN_POSITIVE, 1.0 - NUMERIC_MIN_POSITIVE)
    return eig, safe_timescales(int(max(1, lag)), eig)


# ---------------------------------------------------------------------------
# featurizer geometry (fp32; S/features/deeptica/ts_feature_extractor.py:423-500,
# no PBC) and angle post-processing
# ---------------------------------------------------------------------------
def reference_quirk_its_from_counts(C, lag, n_timescales):
    """What ITSMixin._deterministic_its_from_counts (S/markov_state_model/_its.py:742-801) RETURNS, restated with
    the published deeptime 0.4.5 algorithms it calls (deeptime itself is absent: parity unpinned): "pi" from the
    row sums of T = 1 (:753-754) makes deeptime's dense reversible branch (eigenvalues_rev: S = sqrt(pi) T / sqrt(pi),
    numpy.linalg.eigvalsh(S), decreasing magnitude) run eigvalsh on the non-symmetric T, i.e. on its lower triangle."""
    C = np.asarray(C, dtype=np.float64)
    n = int(n_timescales)
    Crev = 0.5 * (C + C.T)
    row = Crev.sum(axis=1, keepdims=True)
    T = Crev / np.where(row == 0, 1.0, row)
    pi = np.maximum(T.sum(axis=1), 1e-300)
    pi = pi / pi.sum()
    smu = np.sqrt(pi)
    S = smu[:, None] * T / smu
    w = np.linalg.eigvalsh(S)
    w = w[np.argsort(np.abs(w), kind="stable")[::-1]]
    k = T.shape[0]
    k_eval = None if n + 1 > k else n + 1
    wk = w if k_eval is None else w[:k_eval]
    slow = np.sort(wk)[::-1][1:1 + n]
    ev = np.zeros(n)
    ev[:slow.shape[0]] = np.clip(np.abs(slow), NUMERIC_MIN_POSITIVE, 1.0 - NUMERIC_MIN_POSITIVE)
    k_times = None if k_eval is None else min(k, n + 1)
    wt = w if k_times is None else w[:k_times]
    tsr = np.zeros(wt.shape[0])
    one = np.isclose(np.abs(wt), 1.0, rtol=0.0, atol=1e-14)
    tsr[one] = np.inf
    with np.errstate(divide="ignore"):
        tsr[~one] = -float(max(1, int(lag))) / np.log(np.abs(wt[~one]))
    ts = np.full(n, np.nan)
    cut = tsr[1:1 + n]
    ts[:cut.shape[0]] = cut
    with np.errstate(divide="ignore", invalid="ignore"):
        rates = np.where(np.isfinite(ts), 1.0 / ts, np.nan)
    return ev, ts, rates


def distances(xyz, pairs, eps=1e-12):
    xyz = np.asarray(xyz, dtype=np.float32)
    p = np.asarray(pairs)
    v = xyz[:, p[:, 1], :] - xyz[:, p[:, 0], :]
    sq = (v * v).sum(axis=-1, dtype=np.float32)
    return np.sqrt(np.maximum(sq, np.float32(eps)))


def dihedrals(xyz, quads, eps=1e-12):
    xyz = np.asarray(xyz, dtype=np.float32)
    q = np.asarray(quads)
    e = np.float32(eps)
    b0 = xyz[:, q[:, 1]] - xyz[:, q[:, 0]]
    b1 = xyz[:, q[:, 2]] - xyz[:, q[:, 1]]
    b2 = xyz[:, q[:, 3]] - xyz[:, q[:, 2]]
    c0 = np.cross(b0, b1)
    c1 = np.cross(b1, b2)
    n0 = np.sqrt(np.maximum((c0 * c0).sum(-1), e))
    n1 = np.sqrt(np.maximum((c1 * c1).sum(-1), e))
    nb = np.sqrt(np.maximum((b1 * b1).sum(-1), e))
    c0 = c0 / n0[..., None]
    c1 = c1 / n1[..., None]
    b1u = b1 / nb[..., None]
    x = (c0 * c1).sum(-1)
    y = (np.cross(c0, c1) * b1u).sum(-1)
    mask = (np.abs(x) + np.abs(y)) >= e
    ang = np.arctan2(np.where(mask, y, 0.0), np.where(mask, x, 1.0))
    return np.where(mask, ang, 0.0).astype(np.float32)


def angles(xyz, triplets, eps=1e-12):
    xyz = np.asarray(xyz, dtype=np.float32)
    t = np.asarray(triplets)
    e = np.float32(eps)
    v1 = xyz[:, t[:, 0]] - xyz[:, t[:, 1]]
    v2 = xyz[:, t[:, 2]] - xyz[:, t[:, 1]]
    dot = (v1 * v2).sum(-1)
    n1 = np.sqrt(np.maximum((v1 * v1).sum(-1), e))
    n2 = np.sqrt(np.maximum((v2 * v2).sum(-1), e))
    return np.arccos(np.clip(dot / (n1 * n2), -1.0, 1.0)).astype(np.float32)


def wrap_to_minus_pi_pi(a):
    """S/features/builtins.py:11-14."""
    w = ((a + np.pi) % (2 * np.pi)) - np.pi
    return np.where(w <= -np.pi, w + 2 * np.pi, w)


def trig_expand_periodic(X, periodic):
    """S/api/features.py:138-180: periodic column j -> adjacent [cos, sin]."""
    cols, mapping = [], []
    for j in range(X.shape[1]):
        if bool(periodic[j]):
            cols += [np.cos(X[:, j]), np.sin(X[:, j])]
            mapping += [j, j]
        else:
            cols.append(X[:, j])
            mapping.append(j)
    return (np.vstack(cols).T if cols else X), np.asarray(mapping, dtype=int)


# ---------------------------------------------------------------------------
# Chapman-Kolmogorov test (S/validation/ck_rule.py:36-117, S/markov_state_model/ck_runner.py:70-83,
# 155-176, 240-270).  Pinned by tests/golden/ck.npz (ck_rule imported from the reference).
# ---------------------------------------------------------------------------
def ck_error(P_tau, P_k_tau, k):
    """RMS of matrix_power(P_tau, k) - P_k_tau (ck_rule.py:36-47)."""
    P_tau = np.asarray(P_tau, float)
    P_k_tau = np.asarray(P_k_tau, float)
    if P_tau.shape != P_k_tau.shape or P_tau.shape[0] != P_tau.shape[1]:
        raise ValueError("P_tau and P_k_tau must be square matrices of identical shape.")
    d = np.linalg.matrix_power(P_tau, int(k)) - P_k_tau
    return float(np.sqrt(np.mean(d * d)))


def multinomial_rms_se(P, counts):
    """ck_rule.py:50-63: sqrt(mean_i(sum_j p_ij (1 - p_ij) / N_i / n)), bad N_i -> 1."""
    P = np.asarray(P, float)
    counts = np.asarray(counts, float)
    n = P.shape[0]
    if counts.shape[0] != n:
        raise ValueError("counts length must equal number of states.")
    N = np.where(np.isfinite(counts) & (counts > 0.0), counts, 1.0)
    rows = (P * (1.0 - P)).sum(axis=1) / N / n
    return float(np.sqrt(np.mean(rows)))


def decide_ck(P_taus, P_ktaus, row_counts_by_lag, *, mode="ess_adjusted", absolute=0.15, min_pass_fraction=0.8,
              per_lag_cap=0.35, k_steps=(2, 3, 4), sigma_mult=3.0):
    """ck_rule.py:71-117 -> dict(pass_fraction, per_lag, passed)."""
    per_lag, total, passes = {}, 0, 0
    for k in k_steps:
        if k not in P_taus or k not in P_ktaus or k not in row_counts_by_lag:
            continue
        total += 1
        err = ck_error(P_taus[k], P_ktaus[k], k)
        if mode == "absolute":
            thr, noise = absolute, float("nan")
        elif mode == "ess_adjusted":
            noise = multinomial_rms_se(P_ktaus[k], row_counts_by_lag[k])
            thr = float(min(per_lag_cap, sigma_mult * noise))
        else:
            raise ValueError(f"Unknown CK mode: {mode}")
        ok = err <= thr
        passes += int(ok)
        per_lag[k] = {"error": err, "threshold": thr, "noise_rms": noise, "pass": float(ok)}
    frac = passes / total if total else 0.0
    return {"pass_fraction": frac, "per_lag": per_lag, "passed": frac >= min_pass_fraction}


def ck_micro(dtrajs, lag_time, factors=(2, 3, 4, 5), min_trans=50, top_n_micro=50):
    """Microstate branch of run_ck (ck_runner.py:135-153 preprocessing, :240-270 selection, :155-176
    test): states with lag-1 in/out counts > 0 are kept and renumbered (frames in other states are
    DROPPED from the sequences, as the reference does), the top_n_micro most populated (lag tau
    counts) are selected the same way, then mse[f] = mean((T1^f - T_f)^2) for every factor whose
    lag-f*tau count rows all reach min_trans."""
    def count(trajs, n, lag):
        C = np.zeros((n, n))
        for t in trajs:
            t = np.asarray(t, dtype=np.int64)
            if t.size <= lag:
                continue
            a, b = t[:-lag], t[lag:]
            ok = (a >= 0) & (b >= 0) & (a < n) & (b < n)
            np.add.at(C, (a[ok], b[ok]), 1.0)
        return C

    def rownorm(C):
        rs = C.sum(axis=1, keepdims=True)
        rs[rs == 0] = 1.0
        return C / rs

    def relabel(trajs, keep):
        lut = -np.ones(int(max(int(np.max(t)) for t in trajs if len(t)) + 1), dtype=np.int64)
        lut[keep] = np.arange(len(keep))
        return [lut[np.asarray(t, dtype=np.int64)][lut[np.asarray(t, dtype=np.int64)] >= 0] for t in trajs]

    out = {"mse": {}, "insufficient_k": [int(f) for f in factors], "mode": "none"}
    n_states = int(max(int(np.max(t)) for t in dtrajs) + 1)
    C1 = count(dtrajs, n_states, 1)
    active = np.where(C1.sum(axis=1) + C1.sum(axis=0) > 0)[0]
    if active.size == 0:
        return out
    trajs = relabel(dtrajs, active)
    Ctau = count(trajs, active.size, lag_time)
    pops = Ctau.sum(axis=1) + Ctau.sum(axis=0)
    if np.count_nonzero(pops) == 0:
        return out
    top = np.argsort(-pops, kind="stable")[: min(int(top_n_micro), pops.size)]
    micro = relabel(trajs, top)
    n_sel = top.size
    Csel = count(micro, n_sel, lag_time)
    if np.any(Csel.sum(axis=1) < min_trans):
        return out
    T1 = rownorm(Csel)
    for f in factors:
        Ck = count(micro, n_sel, lag_time * int(f))
        if np.any(Ck.sum(axis=1) < min_trans):
            continue
        d = np.linalg.matrix_power(T1, int(f)) - rownorm(Ck)
        out["mse"][int(f)] = float(np.mean(d * d))
        out["insufficient_k"].remove(int(f))
    out["mode"] = "micro"
    out["selected"] = active[top]
    return out


# ---------------------------------------------------------------------------
# Weighted free-energy surface (S/analysis/fes.py:20-88, 91-114, 142-292, 411-599).
# Pinned by tests/golden/fes.npz (compute_weighted_fes imported from the reference).
# ---------------------------------------------------------------------------
KB_KJ_PER_MOL = 0.00831446261815324  # S/constants.py:16


def fes_select_components(coords, n_components=2):
    """fes.py:20-88: the n_components non-constant columns of highest (population) variance."""
    coords = np.asarray(coords)
    if coords.ndim != 2:
        raise ValueError(f"Expected 2D coordinate array, got shape {coords.shape}")
    if coords.shape[1] < n_components:
        raise ValueError(f"Coordinate array has {coords.shape[1]} dimensions, need {n_components}")
    var = np.var(coords, axis=0)
    idx = np.where(var > 0)[0]
    if idx.size < n_components:
        raise ValueError("FES component selection requires at least "
                         f"{n_components} non-constant CV columns; found {idx.size}")
    order = idx[np.argsort(var[idx])[::-1]][:n_components].tolist()
    return coords[:, order], order


def fes_normalise_weights(n_frames, weights):
    """fes.py:91-114 -> (normalised weights, total, effective sample size)."""
    if weights is None:
        return np.full(n_frames, 1.0 / n_frames), float(n_frames), float(n_frames)
    w = np.asarray(weights, dtype=np.float64).reshape(-1)
    if w.shape[0] != n_frames:
        raise ValueError("Frame weights must match number of frames")
    if np.any(w < 0.0) or not np.all(np.isfinite(w)):
        raise ValueError("Frame weights must be finite and non-negative")
    total = float(np.sum(w))
    if total <= 0.0:
        raise ValueError("Frame weights must sum to a positive value")
    return w / total, total, total ** 2 / float(np.sum(w ** 2))


def fes_bandwidth(coord, w_norm, ess, selector):
    """fes.py:142-173 (d = 2)."""
    if isinstance(selector, (int, float)):
        if selector <= 0:
            raise ValueError("Bandwidth must be positive")
        return float(selector)
    mean = float(np.average(coord, weights=w_norm))
    var = float(np.average((coord - mean) ** 2, weights=w_norm))
    if var <= 0.0:
        raise ValueError("Coordinate variance must be positive to compute bandwidth")
    n_eff = max(ess, 1.0)
    sel = str(selector).lower()
    if sel == "scott":
        factor = n_eff ** (-1.0 / 6.0)
    elif sel == "silverman":
        factor = (n_eff * 4.0 / 4.0) ** (-1.0 / 6.0)
    else:
        raise ValueError("Bandwidth must be 'scott', 'silverman', or a positive float")
    return float(np.sqrt(var) * factor)


def fes_smooth_sparse_bins(hist, min_count):
    """fes.py:270-292 with the 3 x 3 'nearest' filter written out."""
    mask = hist < float(min_count)
    if not np.any(mask):
        return hist, 0
    p = np.pad(hist, 1, mode="edge")
    nx, ny = hist.shape
    tot = sum(p[1 + di:1 + di + nx, 1 + dj:1 + dj + ny] for di in (-1, 0, 1) for dj in (-1, 0, 1))
    nm = (tot - hist) / 8.0
    targets = np.maximum(nm, float(min_count))
    upd = mask & (nm > 0.0) & (targets > hist)
    out = hist.copy()
    out[upd] = targets[upd]
    return out, int(np.count_nonzero(upd))


def fes_free_energy(hist, temperature_K):
    """fes.py:570-599."""
    hist = np.asarray(hist, float)
    total = float(np.sum(hist))
    if not np.all(np.isfinite(hist)) or not np.isfinite(total) or total <= 0 or np.any(hist <= 0):
        raise ValueError("Histogram entries must be strictly positive and finite for FES")
    F = -(KB_KJ_PER_MOL * float(temperature_K)) * np.log(hist / total)
    return F - np.min(F)


def weighted_fes(X, *, weights=None, bins=64, temperature_K=300.0, method="kde", bandwidth="scott",
                 min_count_per_bin=1):
    """compute_weighted_fes on a plain (N, d) array (fes.py:411-453 without the dataset plumbing)."""
    coords, sel = fes_select_components(np.asarray(X, float), 2)
    cx, cy = coords[:, 0], coords[:, 1]
    w = None if weights is None else np.asarray(weights, np.float64).reshape(-1)
    meta = {"selected_components": sel}
    if method == "kde":
        w_norm, total, ess = fes_normalise_weights(cx.shape[0], w)
        nx, ny = (int(bins), int(bins)) if np.isscalar(bins) else (int(bins[0]), int(bins[1]))
        bwx, bwy = fes_bandwidth(cx, w_norm, ess, bandwidth), fes_bandwidth(cy, w_norm, ess, bandwidth)
        xe = np.linspace(cx.min() - 3.0 * bwx, cx.max() + 3.0 * bwx, nx + 1)
        ye = np.linspace(cy.min() - 3.0 * bwy, cy.max() + 3.0 * bwy, ny + 1)
        xc, yc = 0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:])
        ex = np.exp(-0.5 * ((xc[:, None] - cx[None, :]) / bwx) ** 2)
        ey = np.exp(-0.5 * ((yc[:, None] - cy[None, :]) / bwy) ** 2)
        hist = np.einsum("ik,jk,k->ij", ex, ey, w_norm) / (2.0 * np.pi * bwx * bwy)
        meta.update(bw_x=bwx, bw_y=bwy, ess=ess, total_weight=total)
    elif method == "grid":
        hist, xe, ye = np.histogram2d(cx, cy, bins=bins, weights=w)
        raw_total = float(hist.sum())
        smoothed = 0
        if min_count_per_bin > 0:
            hist, smoothed = fes_smooth_sparse_bins(hist, int(min_count_per_bin))
            if smoothed > 0 and raw_total > 0 and hist.sum() > 0:
                hist = hist * (raw_total / float(hist.sum()))
        meta["smoothed_bins"] = smoothed
    else:
        raise ValueError("FES method must be either 'kde' or 'grid'")
    return {"histogram": hist, "xedges": xe, "yedges": ye, "free_energy": fes_free_energy(hist, temperature_K),
            "metadata": meta}


# ---------------------------------------------------------------------------
# TPT and lumping (S/markov_state_model/_tpt.py:39-160, 255-347 via deeptime 0.4.5 -- published
# dense algorithm, parity unpinned; _msm_utils.py:103-160 lumping / populations / MFPT).
# ---------------------------------------------------------------------------
def committor(T, source, sink, forward=True, pi=None):
    T = np.asarray(T, float)
    n = T.shape[0]
    A, B = np.unique(np.asarray(source, int)), np.unique(np.asarray(sink, int))
    if forward:
        W = T - np.eye(n)
        r = np.zeros(n)
        r[B] = 1.0
    else:
        pi = stationary_distribution(T) if pi is None else np.asarray(pi, float)
        W = (pi[None, :] * T.T) / pi[:, None] - np.eye(n)
        r = np.zeros(n)
        r[A] = 1.0
    for S in (A, B):
        W[S, :] = 0.0
        W[S, S] = 1.0
    return np.linalg.solve(W, r)


def reactive_flux(T, pi, source, sink):
    T = np.asarray(T, float)
    pi = np.asarray(pi, float)
    qp = committor(T, source, sink, True)
    qm = committor(T, source, sink, False, pi)
    gross = (pi * qm)[:, None] * T * qp[None, :]
    np.fill_diagonal(gross, 0.0)
    net = np.maximum(0.0, gross - gross.T)
    A = np.unique(np.asarray(source, int))
    notA = np.setdiff1d(np.arange(T.shape[0]), A)
    F = float(gross[np.ix_(A, notA)].sum())
    Z = float(np.dot(pi, qm))
    return {"qplus": qp, "qminus": qm, "gross": gross, "net": net, "total_flux": F, "rate": F / Z, "mfpt": Z / F}


def macro_populations(pi_micro, micro_to_macro):
    m = np.asarray(micro_to_macro, int)
    out = np.bincount(m, weights=np.asarray(pi_micro, float), minlength=int(m.max()) + 1)
    s = out.sum()
    return out / s if s > 0 else out


def lump_micro_to_macro_T(T_micro, pi_micro, micro_to_macro):
    m = np.asarray(micro_to_macro, int)
    nM = int(m.max()) + 1
    M = np.zeros((T_micro.shape[0], nM))
    M[np.arange(m.size), m] = 1.0
    F = M.T @ (np.asarray(pi_micro, float)[:, None] * np.asarray(T_micro, float)) @ M
    rows = F.sum(axis=1)
    rows[rows == 0] = 1.0
    return F / rows[:, None]


def macro_mfpt(T_macro):
    T = np.asarray(T_macro, float)
    n = T.shape[0]
    out = np.zeros((n, n))
    for j in range(n):
        mask = np.ones(n, bool)
        mask[j] = False
        try:
            t = np.linalg.solve(np.eye(n - 1) - T[np.ix_(mask, mask)], np.ones(n - 1))
        except np.linalg.LinAlgError:
            t = np.full(n - 1, np.nan)
        out[mask, j] = t
    return out


# ---------------------------------------------------------------------------
# PCA (S/markov_state_model/reduction.py:43-74: _preprocess + sklearn PCA; pinned by tests/golden/pca.npz)
# ---------------------------------------------------------------------------
def pca_reduce(X, n_components=2, scale=True):
    """Covariance-eigh PCA as sklearn >= 1.5 runs it for tall matrices: ddof = 1, descending variance,
    every component's largest-magnitude loading positive (svd_flip on V)."""
    Z = preprocess(X, scale=scale)
    Zc = Z - Z.mean(axis=0)
    C = Zc.T @ Zc / max(Z.shape[0] - 1, 1)
    w, V = np.linalg.eigh(C)
    order = np.argsort(-w, kind="stable")[:n_components]
    comps = V[:, order]
    top = np.argmax(np.abs(comps), axis=0)
    comps = comps * np.sign(comps[top, np.arange(comps.shape[1])])[None, :]
    return Zc @ comps


# ---- regular-grid microstates (S/analysis/discretize.py:517-593) ---------------------------------
class GridStates:
    """_GridDiscretizer restated: per-feature equal-width edges between the training extremes; a
    frame's state is the rank of first appearance of its bin combination, counted over fit() and
    then every transform() call in order."""

    def __init__(self, target_states):
        self.target_states = max(int(target_states), 1)
        self.edges = []
        self.mapping = {}

    def cells(self, X):
        cols = []
        for f, e in enumerate(self.edges):
            cols.append(np.clip(np.digitize(X[:, f], e) - 1, 0, len(e) - 2))
        return np.stack(cols, axis=1)

    def fit(self, X):
        F = X.shape[1]
        bins = max(int(round(self.target_states ** (1.0 / F))), 1)
        self.edges = []
        for f in range(F):
            lo, hi = float(np.min(X[:, f])), float(np.max(X[:, f]))
            if not (np.isfinite(lo) and np.isfinite(hi)):
                raise ValueError("Non-finite values encountered while building grid")
            if lo == hi:
                lo, hi = lo - 0.5, hi + 0.5
            self.edges.append(np.linspace(lo, hi, bins + 1, dtype=np.float64))
        for row in self.cells(X):
            self.mapping.setdefault(tuple(int(v) for v in row), len(self.mapping))
        return self

    def transform(self, X):
        out = np.empty(X.shape[0], dtype=np.int32)
        for i, row in enumerate(self.cells(X)):
            out[i] = self.mapping.setdefault(tuple(int(v) for v in row), len(self.mapping))
        return out

    def centers(self):
        mesh = np.meshgrid(*[(e[:-1] + e[1:]) / 2.0 for e in self.edges], indexing="ij")
        return np.stack([m.ravel() for m in mesh], axis=1)


# ---- posterior transition-matrix samples (engine definition: pmarlo_amd/csrc/posterior.hip) ------
def philox4x32(key, counter):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) on arrays: counter uint32 [..., 4] -> [..., 4]."""
    c = np.array(counter, dtype=np.uint64, copy=True)
    k0 = np.uint64(int(key) & 0xFFFFFFFF)
    k1 = np.uint64((int(key) >> 32) & 0xFFFFFFFF)
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[..., 0]
        p1 = np.uint64(0xCD9E8D57) * c[..., 2]
        n0 = (p1 >> np.uint64(32)) ^ c[..., 1] ^ k0
        n2 = (p0 >> np.uint64(32)) ^ c[..., 3] ^ k1
        c = np.stack([n0, p1 & m32, n2, p0 & m32], axis=-1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return c.astype(np.uint32)


def _unit_open(hi, lo):
    v = ((hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)) >> np.uint64(11)
    return (v.astype(np.float64) + 0.5) * 2.0 ** -53


def sample_transition_matrices(C_active_plus_alpha, seed, first_sample, n_samples):
    """Rows ~ Dirichlet(shape row) from Marsaglia-Tsang gamma variates carried as logarithms, random
    numbers keyed as in the device kernel: counter (column, row, sample, 2*attempt [+1])."""
    A = np.asarray(C_active_plus_alpha, dtype=np.float64)
    n = A.shape[0]
    out = np.empty((n_samples, n, n))
    col, row = np.meshgrid(np.arange(n, dtype=np.uint64), np.arange(n, dtype=np.uint64))
    boost = A < 1.0
    a = np.where(boost, A + 1.0, A)
    d = a - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    for s in range(n_samples):
        lg = np.full((n, n), np.nan)
        todo = A > 0
        attempt = 0
        while todo.any():
            ctr = np.stack([col, row, np.full_like(col, first_sample + s), np.full_like(col, 2 * attempt)], axis=-1)
            r = philox4x32(seed, ctr)
            ctr[..., 3] += np.uint64(1)
            q = philox4x32(seed, ctr)
            u1, u2 = _unit_open(r[..., 0], r[..., 1]), _unit_open(r[..., 2], r[..., 3])
            u3, u4 = _unit_open(q[..., 0], q[..., 1]), _unit_open(q[..., 2], q[..., 3])
            attempt += 1
            x = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
            t = 1.0 + c * x
            with np.errstate(invalid="ignore", divide="ignore"):
                v = t * t * t
                lv = np.log(v)
                ok = (t > 0) & ((u3 < 1.0 - 0.0331 * x ** 4) | (np.log(u3) < 0.5 * x * x + d * (1.0 - v + lv))
                                | (attempt >= 64))
                val = np.log(d) + lv + np.where(boost, np.log(u4) / np.where(boost, A, 1.0), 0.0)
            take = todo & ok
            lg[take] = val[take]
            todo &= ~ok
        lg[~(A > 0)] = -np.inf
        mx = lg.max(axis=1, keepdims=True)
        e = np.exp(lg - mx)
        out[s] = e / e.sum(axis=1, keepdims=True)
    return out


def its_posterior_summary(T_samples, lag, n_timescales, q_low, q_high):
    """_summarize_its_stats (S/markov_state_model/_its.py:543-625) over a stack of sampled matrices."""
    ev = np.stack([its_from_transition_matrix(T, lag, n_timescales)[0] for T in T_samples])
    ts = np.stack([its_from_transition_matrix(T, lag, n_timescales)[1] for T in T_samples])
    with np.errstate(divide="ignore", invalid="ignore"):
        rate = np.where(np.isfinite(ts), 1.0 / ts, np.nan)
    stats = {}
    for name, arr in (("eigenvalues", ev), ("timescales", ts), ("rates", rate)):
        stats[name] = np.nanmedian(arr, axis=0)
        stats[name + "_ci"] = np.stack([np.nanpercentile(arr, q_low, axis=0), np.nanpercentile(arr, q_high, axis=0)], -1)
    return stats


# ---- reversible maximum-likelihood estimate (engine: pmarlo_amd/csrc/revmle.hip) -----------------
def reversible_mle(C, maxerr=1e-8, maxiter=1_000_000):
    """Fixed point x_ij = (c_ij + c_ji) / (c_i/x_i + c_j/x_j) iterated on the row sums (Prinz et al. 2011;
    deeptime's dense reversible estimator as called at S/markov_state_model/_msm_utils.py:250-256).
    Returns (T, pi, iterations)."""
    C = np.asarray(C, dtype=np.float64)
    c = C.sum(axis=1)
    C2 = C + C.T
    x = C2.sum(axis=1)
    x = x / x.sum()
    it = 0
    err = np.inf
    while it < maxiter and err > maxerr:
        with np.errstate(divide="ignore", invalid="ignore"):
            v = np.where(x > 0, c / x, 0.0)
            den = v[:, None] + v[None, :]
            f = np.where((C2 > 0) & (den > 0), C2 / den, 0.0)
        xn = f.sum(axis=1)
        xn = xn / xn.sum()
        mid = 0.5 * (x + xn)
        err = np.max(np.where(mid > 0, np.abs(x - xn) / np.where(mid > 0, mid, 1.0), 0.0))
        x = xn
        it += 1
    v = np.where(x > 0, c / x, 0.0)
    den = v[:, None] + v[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        f = np.where((C2 > 0) & (den > 0), C2 / den, 0.0)
    return f / f.sum(axis=1, keepdims=True), x, it


# ---- PCCA+ (deeptime.markov.pcca as used at S/markov_state_model/_msm_utils.py:284-299) -----------
def pcca_memberships(T, m):
    """Inner simplex start + Roeblitz-Weber crispness optimisation on the m dominant right eigenvectors of a
    reversible T (eigenvectors through the pi-symmetrised matrix, ordered by |eigenvalue|)."""
    from scipy.optimize import fmin

    T = np.asarray(T, dtype=np.float64)
    pi = stationary_distribution(T)
    if not np.allclose(pi[:, None] * T, (pi[:, None] * T).T, rtol=1e-5, atol=1e-15):
        raise ValueError("Transition matrix does not fulfill detailed balance")
    sq = np.sqrt(pi)
    w, U = np.linalg.eigh(sq[:, None] * T / sq[None, :])
    order = np.argsort(-np.abs(w))[:m]
    R = U[:, order] / sq[:, None]
    for q in range(m):
        R[:, q] /= np.sqrt(np.sum(pi * R[:, q] ** 2))
    R[:, 0] = np.abs(R[:, 0])
    # inner simplex algorithm, row by row
    rows = R.copy()
    picked = [max(range(R.shape[0]), key=lambda i: float(np.linalg.norm(rows[i])))]
    rows = rows - R[picked[0]]
    for _ in range(1, m):
        base = rows[picked[-1]].copy()
        if np.linalg.norm(base) > 0:
            base = base / np.linalg.norm(base)
        best, far = -1, -1.0
        for i in range(rows.shape[0]):
            rows[i] = rows[i] - (base @ rows[i]) * base
            d = float(np.linalg.norm(rows[i]))
            if d > far and i not in picked:
                best, far = i, d
        picked.append(best)
    A0 = np.linalg.inv(R[picked])

    def fill(block):
        body = np.hstack([-block.sum(axis=1)[:, None], block])
        top = (-(R[:, 1:] @ body)).max(axis=0)
        A = np.vstack([top[None, :], body])
        return A / top.sum()

    def target(vec):
        A = fill(vec.reshape(m - 1, m - 1))
        return -sum(A[j, i] ** 2 / A[0, i] for i in range(m) for j in range(m))

    A = fill(fmin(target, A0[1:, 1:].ravel(), disp=False).reshape(m - 1, m - 1)) if m > 1 else A0
    chi = np.clip(R @ A, 0.0, 1.0)
    return chi / chi.sum(axis=1, keepdims=True)


# ---- CK / ITS lag selection (S/markov_state_model/ck_its_selector.py:70-599) ---------------------
def _pair_counts(dtrajs, n, lag):
    C = np.zeros((n, n))
    for tr in dtrajs:
        tr = np.asarray(tr)
        if tr.size <= lag:
            continue
        a, b = tr[:-lag], tr[lag:]
        ok = (a >= 0) & (a < n) & (b >= 0) & (b < n)
        np.add.at(C, (a[ok], b[ok]), 1.0)
    return C


def _rownorm_strict(C):
    rs = C.sum(axis=1)
    if rs.min() <= 0:
        raise ValueError("empty row")
    return C / rs[:, None]


def ck_its_evaluate_lag(dtrajs, lag, horizons, n_states, coverage_threshold=0.98, min_median_count=100,
                        diag_mass_threshold=0.6):
    """One candidate lag of select_optimal_lag_ck_its: dict with ck_error, coverage, median, n_macro, passed,
    diag_mass, timescales (slowest first)."""
    from scipy.sparse.csgraph import connected_components

    C = _pair_counts(dtrajs, n_states, lag)
    ncomp, lab = connected_components(((C + C.T) > 0).astype(int), directed=False, return_labels=True)
    coverage = float(np.bincount(lab).max()) / n_states
    tot = C.sum(0) + C.sum(1)
    median = int(np.median(tot[tot > 0])) if np.any(tot > 0) else 0
    out = dict(lag=lag, coverage=coverage, median=median, n_macro=0, ck_error=np.inf, passed=False, diag_mass=np.nan,
               timescales=None)
    if coverage < coverage_threshold or median < min_median_count:
        return out
    try:
        T = _rownorm_strict(C)
        pi = stationary_distribution(T)
        ev = np.sort(np.real(np.linalg.eigvals(T)))[::-1]
        n_cand, width = 2, 0.0
        for m in range(2, min(7, ev.size)):
            if ev[m - 1] - ev[m] > width:
                n_cand, width = m, float(ev[m - 1] - ev[m])
        macro = None
        if n_states > n_cand:
            try:
                chi_soft = pcca_memberships(T, n_cand)
                lab_m = np.argmax(chi_soft, axis=1)
                pops = np.asarray([pi[lab_m == u].sum() for u in np.unique(lab_m)])
                ranks = np.empty(pops.size, dtype=int)
                ranks[np.argsort(-pops, kind="stable")] = np.arange(pops.size)
                macro = ranks[np.searchsorted(np.unique(lab_m), lab_m)]
            except ValueError:
                macro = None
        worst = 0.0
        if macro is not None:
            out["n_macro"] = n_cand
            chi = np.zeros((n_states, n_cand))
            chi[np.arange(n_states), macro] = 1.0
            for h in horizons:
                num = chi.T @ np.diag(pi) @ np.linalg.matrix_power(T, h) @ chi
                den = chi.T @ np.diag(pi) @ chi + np.eye(n_cand) * NUMERIC_MIN_POSITIVE
                pred = num @ np.linalg.inv(den)
                obs = _rownorm_strict(_pair_counts([np.where(np.asarray(t) >= 0, macro[np.asarray(t)], -1) for t in dtrajs],
                                                   n_cand, lag * h))
                worst = max(worst, np.abs(pred - obs).sum() / np.abs(obs).sum())
        else:
            for h in horizons:
                pred = np.linalg.matrix_power(T, h)
                obs = _rownorm_strict(_pair_counts(dtrajs, n_states, lag * h))
                worst = max(worst, np.abs(pred - obs).sum() / np.abs(obs).sum())
        out["ck_error"] = float(worst)
    except Exception:
        return out
    try:
        _, slab = connected_components((C > 0).astype(int), directed=True, connection="strong", return_labels=True)
        act = np.flatnonzero(slab == np.argmax(np.bincount(slab)))
        Tr, _, _ = reversible_mle(C[np.ix_(act, act)])
        w = np.sort(np.abs(np.linalg.eigvals(Tr)))[::-1][1:11]
        out["timescales"] = -lag / np.log(np.clip(w, NUMERIC_MIN_POSITIVE, 1 - NUMERIC_MIN_POSITIVE))
        out["diag_mass"] = float(np.trace(Tr) / act.size)
    except Exception:
        pass
    out["passed"] = bool(np.isfinite(out["diag_mass"]) and out["diag_mass"] >= diag_mass_threshold)
    return out


def ck_its_select(evals, tau_candidates, ck_threshold=0.15):
    for e in sorted(evals, key=lambda e: e["lag"]):
        if e["passed"] and e["ck_error"] <= ck_threshold:
            return e["lag"]
    ok = [e for e in evals if e["passed"]]
    if ok:
        return min(ok, key=lambda e: e["ck_error"])["lag"]
    return min(tau_candidates)


def ck_macro(dtrajs, lag_time, macro_k=4, factors=(2, 3, 4, 5), min_trans=50):
    """Macrostate branch of run_ck (S/markov_state_model/ck_runner.py:178-213): PCCA+ on the lag-1 matrix of
    the connected microstates, trajectories mapped to macrostates, mse[f] = mean((T1^f - T_f)^2).
    Returns None when the branch declines (too few states, gap < 0.01, PCCA+ refuses, thin rows)."""
    n_states = int(max(int(np.max(t)) for t in dtrajs) + 1)
    C1 = _pair_counts(dtrajs, n_states, 1)
    active = np.where(C1.sum(axis=1) + C1.sum(axis=0) > 0)[0]
    lut = -np.ones(n_states, dtype=np.int64)
    lut[active] = np.arange(active.size)
    trajs = [lut[np.asarray(t)][lut[np.asarray(t)] >= 0] for t in dtrajs]
    n = active.size
    if n <= macro_k:
        return None
    T1 = _pair_counts(trajs, n, 1)
    T1 = T1 / np.where(T1.sum(1, keepdims=True) == 0, 1.0, T1.sum(1, keepdims=True))
    ev = np.sort(np.real(np.linalg.eigvals(T1)))[::-1]
    if ev.size <= macro_k or ev[macro_k - 1] - ev[macro_k] < 0.01:
        return None
    try:
        chi = pcca_memberships(T1, macro_k)
    except ValueError:
        return None
    lab = np.argmax(chi, axis=1)
    pi = stationary_distribution(T1)
    uniq = np.unique(lab)
    pops = np.asarray([pi[lab == u].sum() for u in uniq])
    ranks = np.empty(uniq.size, dtype=int)
    ranks[np.argsort(-pops, kind="stable")] = np.arange(uniq.size)
    macro = ranks[np.searchsorted(uniq, lab)]
    m = int(macro.max()) + 1
    mtr = [macro[t] for t in trajs]
    Cm = _pair_counts(mtr, m, lag_time)
    if not np.all(Cm.sum(axis=1) >= min_trans):
        return None
    Tm = Cm / Cm.sum(1, keepdims=True)
    out = {"mse": {}, "insufficient_k": [int(f) for f in factors], "macro": macro, "active": active}
    for f in factors:
        Ck = _pair_counts(mtr, m, lag_time * int(f))
        if np.any(Ck.sum(axis=1) < min_trans):
            continue
        d = np.linalg.matrix_power(Tm, int(f)) - Ck / Ck.sum(1, keepdims=True)
        out["mse"][int(f)] = float(np.mean(d * d))
        out["insufficient_k"].remove(int(f))
    return out


# ---- VAMP reduction (S/markov_state_model/reduction.py:113-148; deeptime VAMP restated) ------------
def vamp_reduce(X, lag=1, n_components=2, scale=True, epsilon=1e-6):
    """_preprocess, then the left singular functions of the whitened cross-covariance
    (Wu & Noe 2020): window means removed, 1/T normalisation, eigenvalues <= epsilon dropped, every
    component's largest-magnitude loading positive.  Returns (Y, singular_values)."""
    Z = preprocess(np.asarray(X), scale=scale)
    A, B = Z[:-lag], Z[lag:]
    T = A.shape[0]
    a0, b0 = A.mean(axis=0), B.mean(axis=0)
    Ac, Bc = A - a0, B - b0
    C00, Ctt, C0t = Ac.T @ Ac / T, Bc.T @ Bc / T, Ac.T @ Bc / T

    def inv_split(C):
        w, V = np.linalg.eigh(0.5 * (C + C.T))
        w, V = w[::-1], V[:, ::-1]
        keep = w > epsilon
        V = V[:, keep]
        for j in range(V.shape[1]):
            V[:, j] *= np.sign(V[np.argmax(np.abs(V[:, j])), j])
        return V / np.sqrt(w[keep])

    L0, Lt = inv_split(C00), inv_split(Ctt)
    U, s, _ = np.linalg.svd(L0.T @ C0t @ Lt)
    dim = min(n_components, L0.shape[1], Lt.shape[1])
    W = L0 @ U[:, :dim]
    for j in range(dim):
        W[:, j] *= np.sign(W[np.argmax(np.abs(W[:, j])), j])
    return (Z - a0) @ W, s


# ---- densities to free energies (S/markov_state_model/free_energy.py:257-414) ----------------------
def free_energy_from_density(density, temperature, mask=None, inpaint=False, tiny=None):
    rho = np.asarray(density, dtype=np.float64)
    floor = np.finfo(np.float64).tiny if tiny is None else float(tiny)
    kT = 1.380649e-23 * float(temperature) * 6.02214076e23 / 1000.0
    F = np.full(rho.shape, np.inf)
    ok = rho > floor
    F[ok] = -kT * np.log(rho[ok])
    if mask is not None and not inpaint:
        F[np.asarray(mask, dtype=bool)] = np.nan
    if np.isfinite(F).any():
        F = F - np.nanmin(F)
    return F


def periodic_kde_2d(theta_x, theta_y, bw=(0.35, 0.35), gridsize=(42, 42)):
    x, y = np.asarray(theta_x, float).ravel(), np.asarray(theta_y, float).ravel()
    gx = np.linspace(-np.pi, np.pi, gridsize[0], endpoint=False)
    gy = np.linspace(-np.pi, np.pi, gridsize[1], endpoint=False)
    wrap = lambda a: np.remainder(a + np.pi, 2.0 * np.pi) - np.pi  # noqa: E731
    ex = np.exp(-0.5 * (wrap(gx[:, None] - x[None, :]) / bw[0]) ** 2)       # [gx, N]
    ey = np.exp(-0.5 * (wrap(gy[:, None] - y[None, :]) / bw[1]) ** 2)       # [gy, N]
    return ex @ ey.T / (x.size * 2.0 * np.pi * bw[0] * bw[1])


def pmf_1d(cv, bins=100, temperature=300.0, periodic=False, range_=None, smoothing_sigma=None):
    from scipy.ndimage import gaussian_filter

    cv = np.asarray(cv, float).ravel()
    rng_ = (float(cv.min()), float(cv.max())) if range_ is None else (float(range_[0]), float(range_[1]))
    H, edges = np.histogram(cv, bins=bins, range=rng_, density=True)
    if smoothing_sigma and smoothing_sigma > 0:
        H = gaussian_filter(H, sigma=float(smoothing_sigma), mode="wrap" if periodic else "reflect")
    return free_energy_from_density(H, temperature), edges, H


# ---- CKMixin (S/markov_state_model/_ck.py:61-358) -------------------------------------------------
def ck_mixin_micro(dtrajs, n_states, lag, factors=(2, 3, 4, 5), max_states=50, min_transitions=5):
    from scipy.sparse.csgraph import connected_components

    def count_T(trajs, n, lg):
        C = _pair_counts(trajs, n, lg)
        rs = C.sum(1)
        rs[rs == 0] = 1.0
        return C / rs[:, None], C

    out = {"mse": {}, "insufficient": False}
    _, C = count_T(dtrajs, n_states, lag)
    _, lab = connected_components(((C + C.T) > 0).astype(int), directed=False, return_labels=True)
    idx = np.where(lab == np.argmax(np.bincount(lab)))[0]
    if idx.size > max_states:
        idx = idx[np.argsort((C + C.T).sum(1)[idx])[::-1]][:max_states]
    lut = -np.ones(n_states, dtype=np.int64)
    lut[idx] = np.arange(idx.size)
    trajs = [lut[np.asarray(t)][lut[np.asarray(t)] >= 0] for t in dtrajs]
    T1, C1 = count_T(trajs, idx.size, lag)
    if np.any(C1.sum(1) < min_transitions):
        out["insufficient"] = True
        return out
    for f in factors:
        Te, Ck = count_T(trajs, idx.size, lag * f)
        if np.any(Ck.sum(1) < min_transitions):
            out["insufficient"] = True
            return out
        d = np.linalg.matrix_power(T1, f) - Te
        out["mse"][f] = float(np.mean(d * d))
    return out


def ck_mixin_select_lag(dtrajs, n_states, taus, factor=2):
    def T_of(lg):
        C = _pair_counts(dtrajs, n_states, lg)
        rs = C.sum(1)
        rs[rs == 0] = 1.0
        return C / rs[:, None]

    mses = []
    for t in taus:
        d = np.linalg.matrix_power(T_of(t), factor) - T_of(t * factor)
        mses.append(float(np.mean(d * d)))
    best = taus[int(np.nanargmin(mses))]
    if best == 1 and 2 in taus and mses[taus.index(2)] <= min(mses) + 1e-12:
        best = 2
    return best, mses


# ---------------------------------------------------------------------------------------------------------
# Structure features (S/features/builtins.py:171-250 call mdtraj; mdtraj is absent from the container, so these
# restate mdtraj 1.10's published algorithms -- parity with mdtraj itself unpinned): Shrake-Rupley
# (geometry/src/sasa.cpp), Baker-Hubbard (geometry/hbond.py) and DSSP (geometry/src/dssp.cpp, a port of DSSP 2.0).
# ---------------------------------------------------------------------------------------------------------
def sasa_sphere_points(n_points):
    """sasa.cpp generate_sphere_points: golden-section spiral, float32."""
    i = np.arange(n_points, dtype=np.float64)
    inc = np.pi * (3.0 - np.sqrt(5.0))
    offset = 2.0 / n_points
    y = i * offset - 1.0 + offset / 2.0
    r = np.sqrt(1.0 - y * y)
    phi = i * inc
    return np.stack([np.cos(phi) * r, y, np.sin(phi) * r], axis=1).astype(np.float32)


def shrake_rupley_atoms(xyz, radii, n_sphere_points=960):
    """Per-atom accessible areas, float32 (n_frames, n_atoms): sasa.cpp asa_frame in fp32 operation order
    (`radii` already include the probe radius)."""
    xyz = np.asarray(xyz, np.float32)
    radii = np.asarray(radii, np.float32)
    pts = sasa_sphere_points(n_sphere_points)
    n, A, _ = xyz.shape
    out = np.zeros((n, A), np.float32)
    const = np.float32(4.0 * np.pi / n_sphere_points)

    def dot3(d):   # (x x + y y) + z z in float32
        return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]

    for t in range(n):
        fr = xyz[t]
        for i in range(A):
            d = fr[i] - fr                                   # r_i - r_j
            cut = radii[i] + radii
            nb = np.nonzero((dot3(d) < cut * cut) & (np.arange(A) != i))[0]
            centered = fr[i] + radii[i] * pts                # float32: multiply, then add
            if len(nb):
                dd = centered[:, None, :] - fr[nb][None, :, :]
                inside = (dot3(dd) < (radii[nb] * radii[nb])[None, :]).any(axis=1)
                n_acc = int((~inside).sum())
            else:
                n_acc = len(pts)
            out[t, i] = const * radii[i] * radii[i] * np.float32(n_acc)
    return out


def baker_hubbard_presence(xyz, triplets, distance_cutoff=0.25, angle_cutoff=2.0 * np.pi / 3.0):
    """hbond.py _compute_bounded_geometry + baker_hubbard: frames in which each triplet meets both criteria."""
    xyz = np.asarray(xyz, np.float32)
    trip = np.asarray(triplets, int).reshape(-1, 3)

    def dist(i, j):
        d = xyz[:, j, :] - xyz[:, i, :]
        return np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)

    a = dist(trip[:, 0], trip[:, 1])     # D - H
    b = dist(trip[:, 1], trip[:, 2])     # H ... A
    c = dist(trip[:, 2], trip[:, 0])     # A ... D
    with np.errstate(invalid="ignore", divide="ignore"):
        cosines = (a * a + b * b - c * c) / (np.float32(2.0) * a * b)
    np.clip(cosines, -1, 1, out=cosines)
    angles = np.arccos(cosines)
    presence = (b < np.float32(distance_cutoff)) & (angles > np.float32(angle_cutoff))
    return presence.sum(axis=0).astype(np.int64)


def dssp_codes(xyz, backbone, chain, proline):
    """Kabsch-Sander assignment per frame and protein residue, uint8 (n_frames, R): 0 loop, 1 H, 2 B, 3 E, 4 G, 5 I,
    6 T, 7 S.  dssp.cpp / DSSP 2.0 structure.cpp step by step (fp32 geometry in Angstrom)."""
    xyz = np.asarray(xyz, np.float32)
    bb = np.asarray(backbone, int).reshape(-1, 4)
    chain = np.asarray(chain, int)
    proline = np.asarray(proline, bool)
    R = bb.shape[0]
    n = xyz.shape[0]
    out = np.zeros((n, R), np.uint8)
    f32 = np.float32

    def dist(a, b):
        d = a - b
        return np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])

    for t in range(n):
        fr = xyz[t] * f32(10.0)
        N, CA, C, O = (fr[bb[:, k]] for k in range(4))
        H = N.copy()
        for r in range(1, R):
            if not proline[r] and chain[r - 1] == chain[r]:
                H[r] = N[r] + (C[r - 1] - O[r - 1]) / dist(C[r - 1], O[r - 1])

        def no_break(a, b):
            for r in range(a, b):
                if chain[r] != chain[r + 1] or dist(C[r], N[r + 1]) > f32(2.5):
                    return False
            return True

        acc = [[-1, -1] for _ in range(R)]
        en = [[f32(0.0), f32(0.0)] for _ in range(R)]

        def energy(donor, acceptor):
            res = f32(0.0)
            if not proline[donor]:
                dHO, dHC = dist(H[donor], O[acceptor]), dist(H[donor], C[acceptor])
                dNC, dNO = dist(N[donor], C[acceptor]), dist(N[donor], O[acceptor])
                if min(dHO, dHC, dNC, dNO) < f32(0.5):
                    res = f32(-9.9)
                else:
                    k = f32(27.888)
                    res = -k / dHO + k / dHC - k / dNC + k / dNO
                res = f32(_roundf(res * f32(1000.0)) / f32(1000.0))   # DSSP compatibility mode
                if res < f32(-9.9):
                    res = f32(-9.9)
            if res < en[donor][0]:
                acc[donor][1], en[donor][1] = acc[donor][0], en[donor][0]
                acc[donor][0], en[donor][0] = acceptor, res
            elif res < en[donor][1]:
                acc[donor][1], en[donor][1] = acceptor, res

        for i in range(R - 1):
            for j in range(i + 1, R):
                if dist(CA[i], CA[j]) < f32(9.0):
                    energy(i, j)
                    if j != i + 1:
                        energy(j, i)

        def bond(d, a):
            return (acc[d][0] == a and en[d][0] < f32(-0.5)) or (acc[d][1] == a and en[d][1] < f32(-0.5))

        ss = [0] * R
        ladders = []   # [i_first, i_last, j_first, j_last, type, alive]
        for i in range(1, R - 4):
            for j in range(i + 3, R - 1):
                typ = 0
                if no_break(i - 1, i + 1) and no_break(j - 1, j + 1):
                    if (bond(i + 1, j) and bond(j, i - 1)) or (bond(j + 1, i) and bond(i, j - 1)):
                        typ = 1
                    elif (bond(i + 1, j - 1) and bond(j + 1, i - 1)) or (bond(j, i) and bond(i, j)):
                        typ = 2
                if not typ:
                    continue
                for lad in ladders:
                    if lad[4] != typ or i != lad[1] + 1:
                        continue
                    if typ == 1 and lad[3] + 1 == j:
                        lad[1], lad[3] = i, j
                        break
                    if typ == 2 and lad[2] - 1 == j:
                        lad[1], lad[2] = i, j
                        break
                else:
                    ladders.append([i, i, j, j, typ, 1])
        for a in range(len(ladders)):
            if not ladders[a][5]:
                continue
            for b in range(a + 1, len(ladders)):
                if not ladders[b][5]:
                    continue
                ibi, iei, jbi, jei = ladders[a][:4]
                ibj, iej, jbj, jej = ladders[b][:4]
                if (ladders[a][4] != ladders[b][4] or not no_break(min(ibi, ibj), max(iei, iej))
                        or not no_break(min(jbi, jbj), max(jei, jej)) or ibj - iei >= 6 or (iei >= ibj and ibi <= iej)):
                    continue
                if ladders[a][4] == 1:
                    bulge = (jbj - jei < 6 and ibj - iei < 3) or (jbj - jei < 3)
                else:
                    bulge = (jbi - jej < 6 and ibj - iei < 3) or (jbi - jej < 3)
                if bulge:
                    ladders[a][0], ladders[a][1] = min(ibi, ibj), max(iei, iej)
                    ladders[a][2], ladders[a][3] = min(jbi, jbj), max(jei, jej)
                    ladders[b][5] = 0
        for lad in ladders:
            if not lad[5]:
                continue
            code = 3 if (lad[1] - lad[0] >= 1 or lad[3] - lad[2] >= 1) else 2
            for r in list(range(lad[0], lad[1] + 1)) + list(range(lad[2], lad[3] + 1)):
                if ss[r] != 3:
                    ss[r] = code
        NONE, START, END, BOTH, MID = 0, 1, 2, 3, 4
        hf = [[NONE] * R for _ in range(3)]
        for s in range(3):
            stride = s + 3
            for i in range(R - stride):
                if no_break(i, i + stride) and bond(i + stride, i):
                    hf[s][i + stride] = BOTH if hf[s][i + stride] == START else END
                    for j in range(i + 1, i + stride):
                        if hf[s][j] == NONE:
                            hf[s][j] = MID
                    hf[s][i] = BOTH if hf[s][i] == END else START
        bend = [False] * R
        for i in range(2, R - 2):
            if not no_break(i - 2, i + 2):
                continue
            u, v = CA[i] - CA[i - 2], CA[i + 2] - CA[i]
            ck = ((u[0] * v[0] + u[1] * v[1]) + u[2] * v[2]) / (np.sqrt((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2])
                                                               * np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
            kappa = np.arccos(np.clip(f32(ck), f32(-1.0), f32(1.0))) * f32(57.29577951308232)
            bend[i] = bool(kappa > f32(70.0))

        def start(r, s):
            return hf[s][r] in (START, BOTH)

        for i in range(1, R - 4):
            if start(i, 1) and start(i - 1, 1):
                for j in range(i, i + 4):
                    ss[j] = 1
        for i in range(1, R - 3):
            if start(i, 0) and start(i - 1, 0) and all(ss[j] in (0, 4) for j in range(i, i + 3)):
                for j in range(i, i + 3):
                    ss[j] = 4
        for i in range(1, R - 5):
            if start(i, 2) and start(i - 1, 2) and all(ss[j] in (0, 5) for j in range(i, i + 5)):
                for j in range(i, i + 5):
                    ss[j] = 5
        for i in range(1, R - 1):
            if ss[i] != 0:
                continue
            turn = any(i >= k and start(i - k, s) for s in range(3) for k in range(1, s + 3))
            if turn:
                ss[i] = 6
            elif bend[i]:
                ss[i] = 7
        out[t] = ss
    return out


def _roundf(x):
    """C roundf: half away from zero, float32."""
    x = np.float32(x)
    return np.float32(np.sign(x) * np.floor(np.abs(x) + np.float32(0.5)))


# ---------------------------------------------------------------------------------------------------------
# k-means++ seeding as the engine defines it (pmarlo_amd/csrc/kmeanspp.hip): integer weights, splitmix64 draws
# ---------------------------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def _kpp_hash(seed: int, j: int) -> int:
    h = ((seed ^ 0x6B2B2B6B6D65616E) + 0x9E3779B97F4A7C15 * (j + 1)) & _M64
    h = ((h ^ (h >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    h = ((h ^ (h >> 27)) * 0x94D049BB133111EB) & _M64
    return h ^ (h >> 31)


def kmeans_plusplus(Y, k: int, seed: int, n_total=None, mean=None, std=None):
    """Indices and coordinates of the k frames the device's k-means++ seeding draws (test infrastructure).

    D^2 with a separate multiply and add per feature in ascending order, integer weights rint(D^2 2^e) with 2^e from
    max |z| (frexp), draw j = floor(h_j W / 2^64) against the cumulative integer weights."""
    Z = np.asarray(Y, np.float64)
    if mean is not None:
        Z = (Z - np.asarray(mean, np.float64)) / np.asarray(std, np.float64)
    n, d = Z.shape
    n_total = float(n if n_total is None else n_total)
    absmax = float(np.nanmax(np.abs(Z))) if Z.size else 0.0
    if not absmax > 0.0:
        absmax = 1.0
    v = (n_total * 4.0 * float(d)) * absmax * absmax
    _, ex = np.frexp(v)
    e = int(np.clip(61 - int(ex), -900, 60))
    scale = float(np.ldexp(1.0, e))
    seed &= _M64
    picked = [(_kpp_hash(seed, 0) * n) >> 64]
    mind = None
    for j in range(1, k):
        c = Z[picked[-1]]
        a = np.zeros(n)
        for f in range(d):
            diff = Z[:, f] - c[f]
            a = a + diff * diff
        mind = a if mind is None else np.where(a < mind, a, mind)
        mind = np.where(np.isnan(mind), 0.0, mind)
        w = np.rint(mind * scale).astype(np.int64)
        cum = np.cumsum(w)
        W = int(cum[-1])
        h = _kpp_hash(seed, j)
        if W <= 0:
            picked.append((h * n) >> 64)
        else:
            r = (h * W) >> 64
            picked.append(int(np.searchsorted(cum, r, side="right")))
    idx = np.asarray(picked, np.int64)
    return idx, Z[idx].copy()
