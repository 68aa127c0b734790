#!/usr/bin/env python3
"""One implied-timescale solve at the bench shape (k = 500 microstates, 5 timescales) on the counts of a bench step,
with the persistent subspace-iteration launch and (MSM_SPEC_PERSIST=0) with one launch pair per iteration; checked
against numpy.  usage: tools/time_spectrum.py [k]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import npport  # noqa: E402
from pmarlo_amd.device import Engine  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    eng = Engine(0)
    rng = np.random.default_rng(0)
    # a metastable chain: 6 blocks, sparse inside (the shape of real count matrices)
    P = np.full((k, k), 1e-4)
    w = k // 6
    for b in range(6):
        P[w * b:w * b + w, w * b:w * b + w] += rng.random((w, w)) ** 4
    P /= P.sum(1, keepdims=True)
    counts = rng.multinomial(2000, P[0], size=1)[0][None, :] * 0 + np.vstack([rng.multinomial(2000, P[i]) for i in range(k)])
    cd = eng.to_device(counts.astype(np.int64))
    tm = eng.transition_matrix(cd, mode=1)
    for rep in range(3):
        eng.sync()
        t0 = time.perf_counter()
        spec = eng.spectrum(tm["T"], n=tm["n_active"], n_its=5, lags=[10.0], allow_unconverged=True)
        eng.sync()
        dt = time.perf_counter() - t0
    ev_ref, ts_ref = npport.its_from_counts(counts, 10, 5)
    ok = np.isfinite(ts_ref)
    err = float(np.max(np.abs(spec["its_ts"][0][ok] - ts_ref[ok]) / ts_ref[ok]))
    print(f"MSM_SPEC_PERSIST={os.environ.get('MSM_SPEC_PERSIST', '1')} k={k}: {dt * 1e3:.3f} ms wall per solve, "
          f"{spec['launches']} launch(es), p={spec['p']}, residual {float(spec['residual'][0]):.2e}, "
          f"its rel err vs numpy {err:.2e}")


if __name__ == "__main__":
    main()
