#!/usr/bin/env python3
"""Per-kernel timings (HIP events on the engine stream) at a BASELINE config."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pmarlo_amd.device import Engine  # noqa: E402
from tests import _gen  # noqa: E402


def timeit(eng, fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    eng.sync()
    ts = []
    for _ in range(reps):
        a, b = eng.event(), eng.event()
        a.record()
        fn()
        b.record()
        ts.append(a.elapsed_ms(b))
    return float(np.median(ts)), float(np.min(ts))


def main():
    n, F, d, k, lag = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1_000_000, 64, 10, 500, 10)))
    eng = Engine(0)
    print(eng.info())
    X = _gen.correlated_series(n, F, seed=1000)
    xd = eng.to_device(X)
    res = {}
    res["moments"] = timeit(eng, lambda: eng.column_moments(xd, ddof=0))
    mean, std, cnt = eng.column_moments(xd, ddof=0)
    inv = eng.to_device(1.0 / std.to_host())
    mom = eng.empty((2 * F * F + 2 * F + 1,), np.float64)
    res["lagged_moments"] = timeit(eng, lambda: eng.lagged_moments(xd, lag, mean, out=mom, assume_finite=True))
    res["tica_solve"] = timeit(eng, lambda: eng.tica_solve(mom, F, scale=std))
    eig, W, m2, rank = eng.tica_solve(mom, F, scale=std)
    rngm = np.random.default_rng(1)
    Bm = rngm.normal(size=(F, F))
    Am = eng.to_device(Bm @ Bm.T / F + np.eye(F))
    res["eigh_random_spd"] = timeit(eng, lambda: eng.eigh(Am))
    print("eigh sweeps", eng.eigh(Am)[2].to_host())
    momh = mom.to_host(); sd = std.to_host()
    C00 = momh[:F*F].reshape(F, F) / (2 * momh[-1]) / np.outer(sd, sd)
    C00d = eng.to_device(C00)
    res["eigh_C00"] = timeit(eng, lambda: eng.eigh(C00d))
    print("eigh C00 sweeps", eng.eigh(C00d)[2].to_host(), "cond", np.linalg.cond(C00))
    Y = eng.empty((n, d), np.float64)
    res["project"] = timeit(eng, lambda: eng.project(xd, mean, inv, W, d, mean2=m2, out=Y))
    Yh = Y.to_host()
    rng = np.random.default_rng(0)
    cen = eng.to_device(Yh[rng.choice(n, k, replace=False)])
    lab = eng.empty((n,), np.int32)
    res["kmeans_assign"] = timeit(eng, lambda: eng.kmeans_assign(Y, cen, labels=lab))
    cnts = eng.empty((k, k), np.int64)
    prs = eng.empty((1,), np.int64)
    res["count_transitions"] = timeit(eng, lambda: eng.count_transitions(lab, k, lag, out=cnts, pairs=prs))
    tm = eng.transition_matrix(cnts, mode=1)
    import time as _t
    eng.sync(); t0 = _t.perf_counter()
    spec = eng.spectrum(tm["T"], n=tm["n_active"], n_its=5, lags=[float(lag)])
    eng.sync(); dt = _t.perf_counter() - t0
    print(f"spectrum (ITS, k={k}): {dt * 1e3:.2f} ms wall, launches {spec['launches']}, residual {spec['residual']}")
    xe = np.linspace(Yh[:, 0].min(), Yh[:, 0].max(), 65)
    ye = np.linspace(Yh[:, 1].min(), Yh[:, 1].max(), 65)
    res["fes_hist2d_64x64"] = timeit(eng, lambda: eng.hist2d(Y, (0, 1), xe, ye))
    xc, yc = 0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:])
    res["fes_kde2d_64x64"] = timeit(eng, lambda: eng.kde2d(Y, (0, 1), xc, yc, 0.1, 0.1, None, 1.0 / n))
    print(f"config n={n} F={F} d={d} k={k} lag={lag}; eigs", eig.to_host()[:4])
    flops = {"lagged_moments": 3 * F * F * n, "kmeans_assign": 2 * k * d * n, "fes_kde2d_64x64": 2 * 64 * 64 * n}
    bytes_ = {"moments": n * F * 4, "project": n * (F * 4 + d * 8), "count_transitions": n * 4,
              "kmeans_assign": n * (d * 8 + 4), "lagged_moments": n * F * 4}
    for name, (med, mn) in res.items():
        extra = ""
        if name in flops:
            extra += f"  {flops[name] / mn / 1e9:8.1f} GFLOP/s"
        if name in bytes_:
            extra += f"  {bytes_[name] / mn / 1e6:8.1f} GB/s"
        print(f"{name:20s} median {med:8.3f} ms  min {mn:8.3f} ms{extra}")


if __name__ == "__main__":
    main()
