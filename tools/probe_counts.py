#!/usr/bin/env python3
"""count_transitions timing vs label statistics (LDS atomic contention probe)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pmarlo_amd.device import Engine  # noqa: E402
from tools.time_kernels import timeit  # noqa: E402

eng = Engine(0)
n, k, lag = 1_000_000, 500, 10
rng = np.random.default_rng(0)
cases = {
    "random": rng.integers(0, k, n).astype(np.int32),
    "constant": np.full(n, 7, np.int32),
    "runs16": np.repeat(rng.integers(0, k, n // 16 + 1), 16)[:n].astype(np.int32),
    "hop4": rng.integers(0, 4, n).astype(np.int32) + 100,
    "slowwalk": (np.cumsum(rng.integers(-1, 2, n)) // 8 % k).astype(np.int32),
}
for name, lab in cases.items():
    d = eng.to_device(lab)
    out = eng.zeros((k, k), np.int64)
    med, mn = timeit(eng, lambda: eng.count_transitions(d, k, lag, out=out))
    ref = np.zeros((k, k), np.int64)
    np.add.at(ref, (lab[:-lag], lab[lag:]), 1)
    ok = np.array_equal(out.to_host(), ref)
    print(f"{name:10s} median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us  exact={ok}")
