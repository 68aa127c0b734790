"""Bayesian ITS scan at the C4 shape (k = 200 microstates, 50 lags, 100 posterior samples per lag):
device time of compute_implied_timescales vs numpy eig on a subset of the same matrices."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pmarlo_amd.device import get_engine  # noqa: E402
from pmarlo_amd.markov_state_model import compute_implied_timescales  # noqa: E402


def chain(k, n, seed):
    rng = np.random.default_rng(seed)
    blocks = 5
    P = np.full((k, k), 0.02 / k)
    w = k // blocks
    for b in range(blocks):
        P[w * b:w * b + w, w * b:w * b + w] += rng.uniform(0.2, 1.0, size=(w, w)) / w
    P /= P.sum(1, keepdims=True)
    cdf = np.cumsum(P, axis=1)
    u = rng.random(n)
    x = np.zeros(n, dtype=np.int32)
    for t in range(1, n):
        x[t] = min(k - 1, int(np.searchsorted(cdf[x[t - 1]], u[t])))
    return x


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    eng = get_engine()
    trajs = [chain(k, 60_000, s) for s in range(4)]
    lags = list(range(1, L + 1))
    for label, ns in (("deterministic", 0), (f"{S} samples/lag", S)):
        compute_implied_timescales(trajs, k, lag_times=lags[:2], n_timescales=5, n_samples=min(ns, 4))  # warm-up
        eng.sync()
        t0 = time.perf_counter()
        res = compute_implied_timescales(trajs, k, lag_times=lags, n_timescales=5, n_samples=ns, random_state=1)
        eng.sync()
        dt = time.perf_counter() - t0
        n_mat = L * max(ns, 1)
        print(f"{label:18s} k={k} L={L}: {dt * 1e3:9.1f} ms  ({n_mat} matrices, {dt / n_mat * 1e6:7.1f} us/matrix)  "
              f"slowest ts at lag {lags[-1]}: {res.timescales[-1, 0]:.2f} [{res.timescales_ci[-1, 0, 0]:.2f}, "
              f"{res.timescales_ci[-1, 0, 1]:.2f}]", flush=True)
    # numpy on the host: stationary vector + eigenvalues of 20 matrices of this order
    rng = np.random.default_rng(0)
    T = rng.random((20, k, k))
    T /= T.sum(-1, keepdims=True)
    t0 = time.perf_counter()
    for M in T:
        np.linalg.eigvals(M)
    dt = (time.perf_counter() - t0) / 20
    print(f"numpy eigvals, one {k}x{k} matrix: {dt * 1e3:.1f} ms  -> {dt * L * S:.1f} s for the scan")


if __name__ == "__main__":
    main()
