"""Host stand-in for pmarlo_amd.device.Engine (TEST INFRASTRUCTURE, backed by the oracle).

`ShardedMSM` calls only Engine methods, so this class lets the very code that runs the N > 1 step on the GPUs
run on CPU ranks: numpy arrays for device arrays, the CPU oracle (oracle/cport.py, oracle/npport.py) and plain
numpy for the kernels, with the SAME buffer layouts and the same fixed-point / hashing rules as the HIP library
(pmarlo_amd/csrc/kmeans.hip, moments.hip).  It is never imported by the product."""

from __future__ import annotations

import numpy as np

from oracle import cport, npport


class HostArray:
    """numpy-backed look-alike of DeviceArray (views share memory like device views do)."""

    def __init__(self, arr: np.ndarray):
        self.a = arr
        self.ptr = arr.ctypes.data

    shape = property(lambda self: self.a.shape)
    dtype = property(lambda self: self.a.dtype)
    size = property(lambda self: int(self.a.size))
    nbytes = property(lambda self: int(self.a.nbytes))

    def to_host(self) -> np.ndarray:
        return self.a.copy()

    def copy_from_host(self, arr):
        self.a[...] = np.asarray(arr, self.a.dtype).reshape(self.a.shape)
        return self

    def zero_(self):
        self.a[...] = 0
        return self

    def fill_bytes_(self, byte: int):
        self.a.view(np.uint8)[...] = int(byte) & 0xFF
        return self

    def copy_from(self, src):
        self.a.reshape(-1).view(np.uint8)[...] = src.a.reshape(-1).view(np.uint8)
        return self

    def view(self, shape, dtype=None, offset_elems: int = 0):
        flat = self.a.reshape(-1)
        dtype = self.a.dtype if dtype is None else np.dtype(dtype)
        n = int(np.prod(shape))
        sub = flat[offset_elems:].view(dtype)[:n] if dtype != self.a.dtype else flat[offset_elems:offset_elems + n]
        return HostArray(sub.reshape(shape))


def _splitmix_u(seed: int, j: int) -> float:
    m = (1 << 64) - 1
    h = (seed + 0x9E3779B97F4A7C15 * (j + 1)) & m
    h = ((h ^ (h >> 30)) * 0xBF58476D1CE4E5B9) & m
    h = ((h ^ (h >> 27)) * 0x94D049BB133111EB) & m
    h ^= h >> 31
    return (h >> 11) * (1.0 / 9007199254740992.0)


class HostEngine:
    def empty(self, shape, dtype):
        return HostArray(np.zeros(shape, dtype))

    zeros = empty

    def to_device(self, arr, dtype=None):
        return HostArray(np.array(arr, dtype=dtype if dtype is not None else np.asarray(arr).dtype, copy=True))

    def kmeans_image_bytes(self, n, d):
        return 0          # the bf16 frame images are a device-side optimisation: labels do not depend on them

    def rcp(self, src, dst):
        dst.a[...] = 1.0 / src.a

    # ---- moments -------------------------------------------------------------------------------------------
    def column_moments_partial(self, x, shift=None, sums=None):
        X = x.a.astype(np.float64)
        first = X[0].copy() if shift is None else shift.a
        return None, HostArray(np.array(first))

    def lagged_moments(self, x, lag, shift, *, assume_finite=False, out=None, symmetric=False, **_):
        F = x.shape[1]
        m = npport.lagged_moments([x.a.astype(np.float64) - shift.a], lag)
        mxy = 0.5 * (m["Mxy_half"] + m["Mxy_half"].T) if symmetric else m["Mxy_half"]
        out.a[...] = np.concatenate([m["Mxx"].ravel(), mxy.ravel(), m["sx"], m["sy"], [float(m["T"])]])
        return out

    def moments_from_lagged(self, x, lag, shift, moments, *, out=None, **_):
        dlt = x.a.astype(np.float64) - shift.a
        out.a[...] = np.concatenate([np.full(x.shape[1], float(x.shape[0])), dlt.sum(0), (dlt ** 2).sum(0)])
        return out

    def standardise_params(self, sums, shift, F, n_rows, with_std=True, out=None):
        mean, scale, inv = out
        cnt, s1, s2 = sums.a[:F], sums.a[F:2 * F], sums.a[2 * F:3 * F]
        mean.a[...] = shift.a + s1 / cnt
        sig = np.sqrt(np.maximum(s2 - s1 * s1 / cnt, 0.0) / n_rows) if with_std else np.ones(F)
        sig = np.where(sig < 10 * np.finfo(float).eps, 1.0, sig)
        scale.a[...] = sig
        inv.a[...] = 1.0 / sig
        return out

    def tica_solve(self, moments, F, *, scale=None, epsilon=1e-6, kinetic_map=True, out=None):
        eig, W, m2, rank = out
        v = moments.a
        m = {"Mxx": v[:F * F].reshape(F, F), "Mxy_half": v[F * F:2 * F * F].reshape(F, F), "sx": v[2 * F * F:2 * F * F + F],
             "sy": v[2 * F * F + F:2 * F * F + 2 * F], "T": v[2 * F * F + 2 * F]}
        model = npport.tica_from_moments(m, epsilon=epsilon, scaling="kinetic_map" if kinetic_map else None,
                                         scale=None if scale is None else scale.a)
        r = model["rank"]
        eig.a[...] = 0.0
        eig.a[:r] = model["eigenvalues"]
        W.a[...] = 0.0
        W.a[:, :r] = model["coefficients"]
        m2.a[...] = model["mean"]
        rank.a[...] = r
        return out

    def project(self, x, mu, inv_sigma, W, d, *, mean2=None, out=None, absmax=None, assume_finite=False):
        z = (x.a.astype(np.float64) - mu.a) * inv_sigma.a
        if mean2 is not None:
            z = z - mean2.a
        out.a[...] = z @ W.a[:, :d]
        if absmax is not None:
            absmax.a[...] = np.abs(out.a).max()
        return out

    # ---- k-means (the rules of pmarlo_amd/csrc/kmeans.hip: seeded stratified start, 2^e fixed point) --------
    def kmeans_fit_begin(self, x, k, *, seed, n_total, tol2, centers=None, state=None, absmax_ready=False, **_):
        Y = x.a.astype(np.float64)
        n = Y.shape[0]
        amax = float(state.a[2]) if absmax_ready else float(np.abs(Y).max())
        if not amax > 0.0:
            amax = 1.0
        e = int(np.clip(61 - int(np.ceil(np.log2(float(n_total) * amax))), -900, 60))
        state.a[...] = [np.ldexp(1.0, e), np.ldexp(1.0, -e), amax, 0.0, tol2, 0.0, 0.0, 0.0]
        for j in range(k):
            t = min(int((j + _splitmix_u(int(seed), j)) * (n / k)), n - 1)
            centers.a[j] = Y[t]
        return centers, state

    def kmeans_accumulate(self, x, centers, state, sums, counts, prev_labels=None, **_):
        if state.a[5] != 0.0:
            return
        Y = x.a.astype(np.float64)
        lab = cport.kmeans_assign(Y, centers.a)
        k, d = centers.shape
        fixed = np.rint(Y * state.a[0]).astype(np.int64)
        if prev_labels is None:
            np.add.at(sums.a.reshape(k, d), lab, fixed)
            counts.a += np.bincount(lab, minlength=k).astype(np.int64)
            return
        # incremental form (msm_kmeans_accumulate_delta): only the frames that changed centre move
        old = prev_labels.a
        moved = np.nonzero(old != lab)[0]
        S = sums.a.reshape(k, d)
        np.add.at(S, lab[moved], fixed[moved])
        np.add.at(counts.a, lab[moved], 1)
        had = moved[old[moved] >= 0]
        np.subtract.at(S, old[had], fixed[had])
        np.subtract.at(counts.a, old[had], 1)
        old[...] = lab

    def kmeans_lloyd_pass(self, x, centers, state, sums, counts, prev_labels=None, **kw):
        self.kmeans_accumulate(x, centers, state, sums, counts, prev_labels=prev_labels, **kw)
        self.kmeans_update(sums, counts, centers, state, clear=False)

    def kmeans_update(self, sums, counts, centers, state, clear=True):
        if state.a[5] != 0.0:
            return
        k, d = centers.shape
        S = sums.a.reshape(k, d)
        shift2 = 0.0
        for j in range(k):
            if counts.a[j] > 0:
                c_new = S[j].astype(np.float64) * state.a[1] / float(counts.a[j])
                shift2 += float(((c_new - centers.a[j]) ** 2).sum())
                centers.a[j] = c_new
        state.a[3] = shift2
        state.a[6] += 1.0
        if shift2 <= state.a[4]:
            state.a[5] = 1.0
        if clear:
            sums.a[...] = 0
            counts.a[...] = 0

    def kmeans_assign(self, x, centers, *, labels=None, **_):
        labels.a[...] = cport.kmeans_assign(x.a.astype(np.float64), centers.a)
        return labels

    # ---- counts / T ------------------------------------------------------------------------------------------
    def count_transitions(self, labels, k, lag, *, out=None, pairs=None, **_):
        c, p = cport.count_transitions(labels.a, k, lag)
        out.a[...] = c
        pairs.a[...] = p
        return out, pairs

    def count_transitions_lagscan(self, labels, k, lags, *, out=None, pairs=None, **_):
        for i, lag in enumerate(lags):
            c, p = cport.count_transitions(labels.a, k, int(lag))
            out.a[i] = c
            pairs.a[i] = p
        return out, pairs

    def row_normalise_into(self, counts, T, rowsum, diag_mass):
        C = counts.a.astype(np.float64)
        T.a[...] = npport.normalise_counts(C)
        rowsum.a[...] = C.sum(1)
        diag_mass.a[...] = np.trace(T.a) / C.shape[0]
