#!/usr/bin/env python3
"""Where the host-to-host operator calls spend their time at the bench shard (1 M x 64 -> TICA 10 -> k = 500):
cProfile of tica_reduce, cluster_microstates and discretize_dataset (blocking ctypes calls show up as their own rows)."""
import cProfile
import io
import pstats
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pmarlo_amd.analysis.discretize import discretize_dataset  # noqa: E402
from pmarlo_amd.markov_state_model.clustering import cluster_microstates  # noqa: E402
from pmarlo_amd.markov_state_model.reduction import tica_reduce  # noqa: E402
from tests import _gen  # noqa: E402


def prof(name, fn, top=22):
    fn()                                   # warm (allocations, first-touch)
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    pr = cProfile.Profile()
    pr.enable()
    out = fn()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(top)
    print(f"==== {name}: {dt * 1e3:.1f} ms (second call)")
    print("\n".join(ln for ln in s.getvalue().splitlines() if ln.strip())[:6000])
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    X = _gen.correlated_series(n, 64, seed=1000)
    Y = prof("tica_reduce", lambda: tica_reduce(X, lag=10, n_components=10))
    prof("cluster_microstates", lambda: cluster_microstates(Y, method="kmeans", n_states=500, random_state=0, max_iter=10,
                                                            tolerance=0.0))
    ds = {"splits": {"train": {"X": Y, "segments": [{"start": 0, "stop": Y.shape[0]}]}}}
    prof("discretize_dataset", lambda: discretize_dataset(ds, cluster_mode="kmeans", n_microstates=500, lag_time=10,
                                                          random_state=0))


if __name__ == "__main__":
    main()
