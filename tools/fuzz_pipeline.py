#!/usr/bin/env python3
"""Random small shapes through the device chain vs the oracle chain (edge-case hunt, not a test)."""
import sys
import traceback
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import cport, npport  # noqa: E402
from pmarlo_amd.device import Engine  # noqa: E402
from pmarlo_amd.pipeline import MSMPipeline  # noqa: E402
from tests import _gen  # noqa: E402

eng = Engine(0)
pipe = MSMPipeline(eng)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for case in range(n_cases):
    F = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 31, 32, 33, 47, 50, 63, 64, 65, 100]))
    n = int(rng.integers(max(40, 3 * F), 5000))
    lag = int(rng.integers(1, 12))
    d = int(rng.integers(1, min(F, 12) + 1))
    k = int(rng.integers(1, min(n // 4, 150) + 1))
    dtype = rng.choice([np.float32, np.float64])
    nseg = int(rng.integers(1, 4))
    cuts = sorted(rng.choice(np.arange(lag + 2, n - lag - 2), size=nseg - 1, replace=False).tolist()) if nseg > 1 else []
    segs = list(zip([0] + cuts, cuts + [n]))
    tag = f"case {case}: n={n} F={F} d={d} k={k} lag={lag} dtype={np.dtype(dtype).name} segs={segs}"
    try:
        X = _gen.correlated_series(n, F, seed=case).astype(dtype)
        if rng.random() < 0.35:                       # missing values: _preprocess imputes the column mean
            X[rng.random(X.shape) < 0.01] = np.nan
            tag += " +nan"
        if rng.random() < 0.3 and F > 2:              # a constant column: sigma -> 1, rank drops
            X[:, int(rng.integers(F))] = 1.25
            tag += " +const"
        res = pipe.run(eng.to_device(X), lag=lag, tica_dim=d, k=k, segments=segs if nseg > 1 else None, seed=case,
                       kmeans_iter=5)
        Xp = npport.preprocess(X.astype(np.float64), scale=True)
        ref = npport.tica_fit([Xp[a:b] for a, b in segs], lag, dim=d)
        rank = int(res.tica.rank.to_host()[0])
        eig = res.tica.eigenvalues.to_host()[:min(d, rank)]
        ok = rank == ref["rank"] and np.allclose(eig[:1], ref["eigenvalues"][:1], rtol=1e-8, atol=1e-10)
        Y = res.projected.to_host()
        lab = res.labels.to_host()
        ok &= np.array_equal(lab, cport.kmeans_assign(Y, res.centers.to_host()))
        C, p = cport.count_transitions(lab, k, lag, segments=segs)
        ok &= np.array_equal(res.counts.to_host(), C) and int(res.extras["pairs"].to_host()[0]) == p
        T = res.transition_matrix.to_host()
        ok &= np.array_equal(T, npport.normalise_counts(C.astype(float)))
        if not ok:
            bad += 1
            print("MISMATCH", tag, "rank", rank, ref["rank"], "eig", eig[:2], ref["eigenvalues"][:2])
    except Exception as exc:
        bad += 1
        print("EXCEPTION", tag, repr(exc))
        traceback.print_exc(limit=2)
print(f"{n_cases} cases, {bad} bad")
