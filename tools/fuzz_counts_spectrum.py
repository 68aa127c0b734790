#!/usr/bin/env python3
"""Edge-case hunt (not a test): random shapes through the transition-count paths against the C oracle (bit-exact), and
random stochastic matrices through msm_spectrum (powered and plain iterations) against numpy.
  tools/fuzz_counts_spectrum.py [seed] [n_cases]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import cport  # noqa: E402
from pmarlo_amd.device import Engine  # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for case in range(n_cases):
    k = int(rng.choice([2, 3, 7, 16, 31, 64, 100, 255, 256, 257, 500, 777, 1200, 2048, 2100]))
    n = int(rng.integers(max(2 * k, 50), max(4 * k * k, 5000)))
    n = min(n, 6_000_000)
    lag = int(rng.integers(1, 60))
    stride = int(rng.choice([1, 1, 1, 2, 5]))
    kind = rng.choice(["random", "runs", "walk", "few"])
    if kind == "random":
        lab = rng.integers(0, k, n)
    elif kind == "runs":
        lab = np.repeat(rng.integers(0, k, n // 23 + 1), 23)[:n]
    elif kind == "walk":
        lab = (np.cumsum(rng.integers(-1, 2, n)) // 5) % k
    else:
        lab = rng.choice(rng.integers(0, k, 3), n)
    lab = lab.astype(np.int32)
    if rng.random() < 0.5:
        lab[rng.random(n) < 0.001] = -1
        lab[rng.random(n) < 0.0005] = k + 3
    nseg = int(rng.integers(1, 5))
    cuts = sorted(rng.choice(np.arange(1, n), size=nseg - 1, replace=False).tolist()) if nseg > 1 else []
    segs = list(zip([0] + cuts, cuts + [n]))
    tag = f"counts case {case}: n={n} k={k} lag={lag} stride={stride} kind={kind} segs={len(segs)}"
    try:
        want, pw = cport.count_transitions(lab, k, lag, segments=segs, stride=stride)
        s = np.asarray([a for a, _ in segs], np.int64)
        e = np.asarray([b for _, b in segs], np.int64)
        c, p = eng.count_transitions(eng.to_device(lab), k, lag, starts=s, stops=e, stride=stride)
        if not np.array_equal(c.to_host(), want) or int(p.to_host()[0]) != pw:
            bad += 1
            print("MISMATCH", tag)
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(ex)[:200])
print(f"counts: {n_cases} cases, {bad} bad")

bad2 = 0
for case in range(n_cases):
    k = int(rng.choice([3, 5, 8, 12, 13, 20, 33, 64, 150, 300]))
    nb = int(rng.integers(2, 6))
    P = np.full((k, k), 1e-3 * rng.random())
    w = max(1, k // nb)
    for b in range(nb):
        m = min(w, k - w * b)
        if m <= 0:
            break
        P[w * b:w * b + m, w * b:w * b + m] += rng.random((m, m)) ** rng.integers(1, 5)
    if rng.random() < 0.4:                      # a drift between the blocks: complex pairs
        P += 0.05 * rng.random() * np.roll(np.eye(k), w, axis=1)
    P /= P.sum(1, keepdims=True)
    n_its = int(min(k - 1, rng.integers(1, 7)))
    ev = np.linalg.eigvals(P)
    ev = ev[np.argsort(-np.abs(ev))]
    ref = np.sort(np.abs(np.sort(ev[:n_its + 1].real)[::-1][1:]))[::-1] if False else None
    tag = f"spectrum case {case}: k={k} blocks={nb} n_its={n_its}"
    try:
        a = eng.spectrum(eng.to_device(P), n_its=n_its, lags=[1.0], squarings=2, allow_unconverged=True)
        b = eng.spectrum(eng.to_device(P), n_its=n_its, lags=[1.0], squarings=0, allow_unconverged=True)
        ra, rb = a["ritz"][0][:n_its + 1], b["ritz"][0][:n_its + 1]
        # the leading Ritz values as multisets against numpy (order inside a group of equal moduli is free)
        key = lambda z: np.sort_complex(np.round(z, 7))  # noqa: E731
        conv = float(a["residual"][0]) <= 1e-9 and float(b["residual"][0]) <= 1e-9
        if conv and (np.max(np.abs(key(ra) - key(ev[:n_its + 1]))) > 1e-6 or np.max(np.abs(key(rb) - key(ev[:n_its + 1]))) > 1e-6):
            gapped = abs(abs(ev[n_its]) - abs(ev[n_its + 1])) > 1e-6 if k > n_its + 1 else True
            if gapped:
                bad2 += 1
                print("MISMATCH", tag, ra, ev[:n_its + 1])
    except Exception as ex:  # noqa: BLE001
        bad2 += 1
        print("ERROR", tag, repr(ex)[:200])
print(f"spectrum: {n_cases} cases, {bad2} bad")
