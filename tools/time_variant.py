#!/usr/bin/env python3
"""Time one stage with a variant build of the library: tools/time_variant.py LIB STAGE [n F lag]  (STAGE: cov | tica | project | filter).
The variant replaces pmarlo_amd/csrc/libmsmhip.so for this process only (tools/build_variant.sh makes them)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import pmarlo_amd._lib as _lib  # noqa: E402

if sys.argv[1] != "-":
    _lib.LIB_PATH = (ROOT / sys.argv[1]).resolve()
from pmarlo_amd.device import Engine  # noqa: E402
from tests import _gen  # noqa: E402


def timeit(eng, fn, reps=20, warm=3):
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
    stage = sys.argv[2]
    n, F, lag = (int(v) for v in (sys.argv[3:6] if len(sys.argv) > 5 and sys.argv[3].isdigit() else (1_000_000, 64, 10)))
    eng = Engine(0)
    X = _gen.correlated_series(n, F, seed=1000)
    xd = eng.to_device(X)
    mean, std, cnt = eng.column_moments(xd, ddof=0)
    mom = eng.empty((2 * F * F + 2 * F + 1,), np.float64)
    if stage in ("cov", "covsym"):
        med, mn = timeit(eng, lambda: eng.lagged_moments(xd, lag, mean, out=mom, assume_finite=True, symmetric=stage == "covsym"))
    elif stage == "tica":
        eng.lagged_moments(xd, lag, mean, out=mom, assume_finite=True)
        med, mn = timeit(eng, lambda: eng.tica_solve(mom, F, scale=std))
    elif stage == "kmeans":   # assign / accumulate passes at the bench shard (wrong results are fine for diagnostic variants)
        from pmarlo_amd.dist import ShardConfig, ShardedMSM
        d, k = 10, 500
        rng = np.random.default_rng(0)
        Y = eng.to_device(rng.normal(size=(n, d)))
        cen = eng.to_device(rng.normal(size=(k, d)))
        lab = eng.empty((n,), np.int32)
        img = eng.kmeans_pack(Y)
        med, mn = timeit(eng, lambda: eng.kmeans_assign(Y, cen, labels=lab, image=img))
        print(f"{sys.argv[1]:48s} kmeans assign: median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us")
        if sys.argv[1] != "-" and "full" not in sys.argv:
            return   # diagnostic variants: the assign pass only (append `full` for the accumulate passes)
        # accumulate passes: full sums, then delta sums against labels that barely move
        cfg = ShardConfig(n_frames=n, n_features=F, tica_dim=d, k=k, lag=lag, kmeans_iters=10, seed=0, n_total=n)
        msm = ShardedMSM(eng, cfg, xd)
        msm.step()
        st = msm.buf["fit_state"]
        sums, counts = eng.zeros((k * d,), np.int64), eng.zeros((k,), np.int64)
        img2 = eng.kmeans_pack(msm.Y)
        med, mn = timeit(eng, lambda: eng.kmeans_accumulate(msm.Y, msm.buf["centers"], st, sums, counts, image=img2))
        print(f"{sys.argv[1]:48s} kmeans accumulate (full): median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us")
        prev = eng.empty((n,), np.int32)
        prev.fill_bytes_(0xFF)
        eng.kmeans_accumulate(msm.Y, msm.buf["centers"], st, sums, counts, image=img2, prev_labels=prev)
        med, mn = timeit(eng, lambda: eng.kmeans_accumulate(msm.Y, msm.buf["centers"], st, sums, counts, image=img2, prev_labels=prev))
        print(f"{sys.argv[1]:48s} kmeans accumulate (delta, nothing moves): median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us")
        return
    elif stage == "counts":   # lag-tau transition counts of Markov-chain labels, k = 500
        k = 500
        lab = eng.to_device(_gen.markov_labels(n, k, 3))
        eng.count_transitions(lab, k, lag)
        med, mn = timeit(eng, lambda: eng.count_transitions(lab, k, lag))
        print(f"{sys.argv[1]:48s} counts: median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us")
        return
    else:
        raise SystemExit("stage?")
    print(f"{sys.argv[1]:48s} {stage}: median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us  checksum {float(np.abs(mom.to_host()).sum()):.17g}")


main()
