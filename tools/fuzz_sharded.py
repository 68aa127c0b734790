#!/usr/bin/env python3
"""Random shapes through the fused sharded step (pmarlo_amd/dist.ShardedMSM, one rank) vs the oracle
chain: standardisation parameters, TICA eigenvalues, labels given the device's own projection and centres,
transition counts (edge-case hunt for the fused moments / max-abs passes, not a test)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import cport, npport  # noqa: E402
from pmarlo_amd.device import Engine  # noqa: E402
from pmarlo_amd.dist import ShardConfig, ShardedMSM  # noqa: E402
from tests import _gen  # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for case in range(n_cases):
    F = int(rng.choice([3, 4, 8, 16, 17, 32, 48, 64, 100, 128]))
    n = int(rng.integers(max(400, 6 * F), 60_000))
    lag = int(rng.integers(1, 25))
    d = int(rng.integers(1, min(F, 12) + 1))
    k = int(rng.integers(2, min(n // 8, 200) + 1))
    dtype = rng.choice([np.float32, np.float64])
    tag = f"case {case}: n={n} F={F} d={d} k={k} lag={lag} dtype={np.dtype(dtype).name}"
    try:
        X = (_gen.correlated_series(n, F, seed=int(rng.integers(1 << 30))).astype(np.float64)
             * rng.uniform(0.2, 4.0, size=F) + rng.normal(size=F) * rng.choice([0.0, 1.0, 50.0])).astype(dtype)
        cfg = ShardConfig(n_frames=n, n_features=F, tica_dim=d, k=k, lag=lag, kmeans_iters=3, seed=case)
        msm = ShardedMSM(eng, cfg, eng.to_device(X))
        msm.step()
        msm.step()          # second step: buffers reused
        eng.sync()
        Xd = X.astype(np.float64)
        mean, scale = msm.mean.to_host(), msm.scale.to_host()
        sd = Xd.std(0)
        ok = np.allclose(mean, Xd.mean(0), rtol=1e-10, atol=1e-10 * (np.abs(Xd).max() + 1))
        ok &= np.allclose(scale, np.where(sd < 10 * np.finfo(float).eps, 1.0, sd), rtol=1e-8)
        ref = npport.tica_fit([npport.preprocess(Xd, scale=True)], lag, dim=d)
        rank = int(msm.rank_d.to_host()[0])
        got = msm.eig.to_host()[:min(d, rank, int(ref["rank"]))]
        want = ref["eigenvalues"][:got.size]
        big = np.abs(want) > 1e-3
        ok &= rank == int(ref["rank"]) and np.allclose(got[big], want[big], rtol=1e-7)
        Y, C = msm.Y.to_host(), msm.buf["centers"].to_host()
        labels = msm.labels.to_host()
        ok &= np.array_equal(labels, cport.kmeans_assign(Y, C))
        counts = msm.buf["counts"].view((k, k)).to_host()
        wc, wp = cport.count_transitions(labels, k, lag)
        ok &= np.array_equal(counts, wc) and int(msm.buf["counts"].view((1,), offset_elems=k * k).to_host()[0]) == wp
        amax = float(msm.buf["fit_state"].to_host()[2])
        ok &= amax == np.abs(Y).max()
        if not ok:
            bad += 1
            print("MISMATCH", tag, "rank", rank, ref["rank"], got, want, flush=True)
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(exc), flush=True)
print(f"{n_cases} cases, {bad} bad")
