#!/usr/bin/env python3
"""k-means passes at the bench shard (C3: 1 M x 10 projected coordinates, k = 500): frame-image build, filter
assign / accumulate with a prebuilt image (full sums and the delta mode the bench step runs), the plain entry
points, the whole step, and the share of frames that took the exhaustive scan.
  tools/time_kmeans_filter.py [--lib PATH] [n]      (--lib: a variant build from tools/build_variant.sh)
MSM_KMEANS_FILTER=0 in the environment times the all-fp64 kernel instead."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
args = sys.argv[1:]
lib_name = "-"
if args and args[0] == "--lib":
    import pmarlo_amd._lib as _lib

    lib_name = args[1]
    if lib_name != "-":
        _lib.LIB_PATH = (ROOT / lib_name).resolve()
    args = args[2:]
from pmarlo_amd.device import Engine  # noqa: E402
from pmarlo_amd.dist import ShardConfig, ShardedMSM  # noqa: E402
from tests import _gen  # noqa: E402
from tools.time_kernels import timeit  # noqa: E402


def main():
    n = int(args[0]) if args else 1_000_000
    F, d, k, lag = 64, 10, 500, 10
    eng = Engine(0)
    cfg = ShardConfig(n_frames=n, n_features=F, tica_dim=d, k=k, lag=lag, kmeans_iters=10, seed=0, n_total=n)
    msm = ShardedMSM(eng, cfg, eng.to_device(_gen.correlated_series(n, F, seed=1000)))
    msm.step()
    eng.sync()
    Y, cen, st = msm.Y, msm.buf["centers"], msm.buf["fit_state"]
    lab = eng.empty((n,), np.int32)
    sums, counts = eng.zeros((k * d,), np.int64), eng.zeros((k,), np.int64)
    print("lib", lib_name, "filter", os.environ.get("MSM_KMEANS_FILTER", "1"), "n", n)
    img = eng.kmeans_pack(Y)
    res = {}
    if img is not None:
        res["pack"] = timeit(eng, lambda: eng.kmeans_pack(Y, image=img))
    eng.kmeans_filter_scanned(reset=True)
    res["assign (image given)"] = timeit(eng, lambda: eng.kmeans_assign(Y, cen, labels=lab, image=img), reps=20)
    scanned = eng.kmeans_filter_scanned(reset=True)
    print(f"scanned per pass: {scanned / 22:.0f} of {n} frames ({scanned / 22 / n * 100:.3f} %)")
    res["accumulate (full sums)"] = timeit(
        eng, lambda: eng.kmeans_accumulate(Y, cen, st, sums, counts, image=img), reps=20)
    # the pass of the bench step: delta sums against labels that no longer move (centres fixed)
    prev = eng.empty((n,), np.int32)
    prev.fill_bytes_(0xFF)
    eng.kmeans_accumulate(Y, cen, st, sums, counts, image=img, prev_labels=prev)
    res["accumulate (delta, steady)"] = timeit(
        eng, lambda: eng.kmeans_accumulate(Y, cen, st, sums, counts, image=img, prev_labels=prev), reps=20)
    res["assign (plain call)"] = timeit(eng, lambda: eng.kmeans_assign(Y, cen, labels=lab))
    res["step"] = timeit(eng, msm.step, reps=10)
    # the accumulate launches inside the step (HIP events around each, as bench.py takes them)
    msm.time_accum = True
    msm.accum_events.clear()
    for _ in range(5):
        msm.step()
    eng.sync()
    acc = [a.elapsed_ms(b) for a, b in msm.accum_events]
    msm.time_accum = False
    flops = 2.0 * k * d * n
    for name, (med, mn) in res.items():
        extra = f"  {flops / med / 1e9:8.1f} GFLOP/s algorithmic" if "assign" in name or "accumulate" in name else ""
        print(f"{name:28s} median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us{extra}")
    print(f"{'accumulate inside the step':28s} mean   {np.mean(acc) * 1e3:8.1f} us  min {np.min(acc) * 1e3:8.1f} us  "
          f"first-of-step mean {np.mean(acc[0::10]) * 1e3:8.1f} us  last-of-step mean {np.mean(acc[9::10]) * 1e3:8.1f} us")


if __name__ == "__main__":
    main()
