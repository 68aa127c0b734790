#!/usr/bin/env python3
"""k-means assign / accumulate throughput for wide frames (C5: d = 256, k = 2000) and mid widths."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pmarlo_amd.device import Engine  # noqa: E402
from tools.time_kernels import timeit  # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(0)
for n, d, k in ((1_000_000, 10, 2000), (1_000_000, 4, 100), (200_000, 256, 2000), (400_000, 64, 1000), (500_000, 32, 500), (1_000_000, 16, 500), (1_000_000, 10, 500)):
    X = rng.normal(size=(n, d)).astype(np.float32)
    xd = eng.to_device(X)
    cen = eng.to_device(X[rng.choice(n, k, replace=False)].astype(np.float64))
    lab = eng.empty((n,), np.int32)
    med, mn = timeit(eng, lambda: eng.kmeans_assign(xd, cen, labels=lab), reps=5, warm=1)
    print(f"assign n={n} d={d} k={k}: {mn:8.3f} ms  {2.0 * n * k * d / mn / 1e9:7.1f} TFLOP/s")
