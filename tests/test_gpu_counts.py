"""HIP transition-count kernels vs the oracle (bit-exact) -- through the C ABI."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import cport
from tests import _gen

pytestmark = pytest.mark.gpu


def _bounds(segs):
    return np.asarray([a for a, _ in segs], np.int64), np.asarray([b for _, b in segs], np.int64)


@pytest.mark.parametrize("name,lag,use_segs,stride", [
    ("plain", 3, False, 1), ("segs", 5, True, 1), ("stride", 4, True, 3), ("lag_ge_len", 1300, True, 1)])
def test_counts_golden(engine, golden, name, lag, use_segs, stride):
    g = golden("counts.npz")
    lab = engine.to_device(g["labels"], np.int32)
    kw = {}
    if use_segs:
        kw["starts"], kw["stops"] = _bounds(g["segments"])
    c, p = engine.count_transitions(lab, 7, lag, stride=stride, **kw)
    np.testing.assert_array_equal(c.to_host(), g[f"{name}_counts"].astype(np.int64))
    assert int(p.to_host()[0]) == int(g[f"{name}_pairs"])


def test_two_well_golden(engine, golden):
    g = golden("counts.npz")
    lab = engine.to_device(g["two_well_labels"].astype(np.int32))
    c, p = engine.count_transitions(lab, 2, 400)
    np.testing.assert_array_equal(c.to_host(), g["two_well_counts"].astype(np.int64))
    assert int(p.to_host()[0]) == int(g["two_well_total_pairs"])


def test_weighted_golden(engine, golden):
    g = golden("counts.npz")
    lab = engine.to_device(g["labels"], np.int32)
    w = engine.to_device(g["weights"], np.float64)
    s, e = _bounds(g["segments"])
    c, p = engine.count_transitions_weighted(lab, w, 7, 2, starts=s, stops=e)
    # fp64 atomics: equal to the reference up to summation order
    np.testing.assert_allclose(c.to_host(), g["weighted_counts"], rtol=1e-12)
    assert int(p.to_host()[0]) == int(g["weighted_pairs"])


def test_weighted_integer_weights_exact(engine):
    rng = np.random.default_rng(3)
    lab_h = rng.integers(0, 30, 200_000).astype(np.int32)
    w_h = rng.integers(1, 8, lab_h.size).astype(np.float64)
    want, pw = cport.count_transitions(lab_h, 30, 7, weights=w_h)
    c, p = engine.count_transitions_weighted(engine.to_device(lab_h), engine.to_device(w_h), 30, 7)
    np.testing.assert_array_equal(c.to_host(), want)
    assert int(p.to_host()[0]) == pw


@pytest.mark.parametrize("n,k,lag", [
    (1_000_000, 500, 10),     # C3: row-blocked LDS path
    (100_000, 100, 10),       # C2: single row block
    (300_000, 200, 1),        # C4
    (400_000, 2000, 10),      # C5: k too large for LDS rows -> global atomics
    (1000, 3, 999), (1000, 3, 1000), (17, 20, 1), (1, 2, 1), (0, 4, 1)])
def test_counts_vs_oracle_sizes(engine, n, k, lag):
    lab_h = _gen.markov_labels(n, k, seed=n % 97 + k) if n else np.zeros(0, np.int32)
    if n > 100:
        lab_h[::5003] = -1          # invalid frames (reference: skipped)
        lab_h[7::9001] = k + 5      # out-of-range labels are skipped, never written
    want, pw = cport.count_transitions(lab_h, k, lag)
    c, p = engine.count_transitions(engine.to_device(lab_h, np.int32), k, lag)
    np.testing.assert_array_equal(c.to_host(), want)
    assert int(p.to_host()[0]) == pw
    v = engine.state_counts(engine.to_device(lab_h, np.int32), k)
    np.testing.assert_array_equal(v.to_host(), cport.state_counts(lab_h, k))


@pytest.mark.parametrize("kind", ["constant", "runs", "two_states", "random"])
@pytest.mark.parametrize("n,k,lag", [(300_000, 257, 7), (2_000_000, 1200, 3), (70_000, 256, 1)])
def test_bucket_path_label_statistics(engine, kind, n, k, lag):
    """The two-pass bucket path (pairs >= bins): rows per bucket that do not divide k, several bin copies or one,
    long dwells (run combining) and labels that all land in one bucket."""
    rng = np.random.default_rng(k + lag)
    if kind == "constant":
        lab_h = np.full(n, k - 1, np.int32)
    elif kind == "runs":
        lab_h = np.repeat(rng.integers(0, k, n // 37 + 1), 37)[:n].astype(np.int32)
    elif kind == "two_states":
        lab_h = (rng.integers(0, 2, n) * (k - 1)).astype(np.int32)
    else:
        lab_h = rng.integers(0, k, n).astype(np.int32)
    lab_h[::4099] = -1
    lab_h[3::7001] = k
    want, pw = cport.count_transitions(lab_h, k, lag)
    c, p = engine.count_transitions(engine.to_device(lab_h), k, lag)
    np.testing.assert_array_equal(c.to_host(), want)
    assert int(p.to_host()[0]) == pw


def test_bucket_path_many_chunks(engine, monkeypatch):
    """More first-pass workgroups than one tile of the second pass gathers (1024): forced with tiny chunks."""
    monkeypatch.setenv("MSM_COUNTS_CHUNK", "64")
    n, k = 100_000, 120
    lab_h = _gen.markov_labels(n, k, 3)
    segs = [(0, 30_001), (30_001, 30_002), (30_010, n)]
    s, e = _bounds(segs)
    for lag, stride in [(9, 1), (2, 3)]:
        want, pw = cport.count_transitions(lab_h, k, lag, segments=segs, stride=stride)
        c, p = engine.count_transitions(engine.to_device(lab_h), k, lag, starts=s, stops=e, stride=stride)
        np.testing.assert_array_equal(c.to_host(), want)
        assert int(p.to_host()[0]) == pw


def test_many_segments_and_ragged(engine):
    rng = np.random.default_rng(0)
    n, k = 250_000, 64
    lab_h = _gen.markov_labels(n, k, 5)
    cuts = np.sort(rng.choice(np.arange(1, n), size=99, replace=False))
    edges = np.concatenate([[0], cuts, [n]])
    segs = [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:])]
    segs += [(n - 5, n + 50), (-10, 3)]  # clipped like _iter_segments
    for lag, stride in [(25, 1), (3, 4)]:
        want, pw = cport.count_transitions(lab_h, k, lag, segments=segs, stride=stride)
        s, e = _bounds(segs)
        c, p = engine.count_transitions(engine.to_device(lab_h), k, lag, starts=s, stops=e, stride=stride)
        np.testing.assert_array_equal(c.to_host(), want)
        assert int(p.to_host()[0]) == pw


def test_lagscan_matches_per_lag(engine):
    n, k = 200_000, 200
    lab_h = _gen.markov_labels(n, k, 11)
    lags = list(range(1, 51))
    segs = [(0, 80_000), (80_000, 200_000)]
    s, e = _bounds(segs)
    c, p = engine.count_transitions_lagscan(engine.to_device(lab_h), k, lags, starts=s, stops=e)
    ch, ph = c.to_host(), p.to_host()
    for i in (0, 9, 49):
        want, pw = cport.count_transitions(lab_h, k, lags[i], segments=segs)
        np.testing.assert_array_equal(ch[i], want)
        assert int(ph[i]) == pw
    # size-independent property: every lag counts exactly expected_pairs pairs
    for i, lag in enumerate(lags):
        assert int(ch[i].sum()) == int(ph[i]) == sum(max(0, (b - a) - lag) for a, b in segs)


def test_invalid_arguments_raise(engine):
    lab = engine.to_device(np.zeros(10, np.int32))
    with pytest.raises(ValueError):
        engine.count_transitions(lab, 3, 0)
    with pytest.raises(ValueError):
        engine.count_transitions(lab, 0, 1)
