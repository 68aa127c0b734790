"""compute_analysis_debug on the GPU (counts, state visits, dwell-time runs from device kernels) against the
summaries the reference's analysis/debug_export.py produced for the same label sequences
(tests/golden/debug.json / debug.npz, made by importing the reference).  Everything is integer or a ratio of
integers: compared exactly, floats to 1e-13."""
import json

import numpy as np
import pytest

from pmarlo_amd.analysis.debug_export import analyse_scc, compute_analysis_debug
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _same(a, b, path=""):
    if isinstance(b, dict):
        assert isinstance(a, dict) and sorted(a) == sorted(b), path
        for k in b:
            _same(a[k], b[k], f"{path}/{k}")
    elif isinstance(b, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    elif isinstance(b, float):
        assert a == pytest.approx(b, rel=1e-13, abs=1e-15, nan_ok=True), path
    else:
        assert a == b, (path, a, b)


@pytest.mark.parametrize("case", ["sliding", "strided", "tiny"])
def test_compute_analysis_debug_vs_reference_golden(golden, case):
    want = json.loads((GOLDEN / "debug.json").read_text())[case]
    g = golden("debug.npz")
    dtrajs = [g[f"{case}_dtraj{i}"] for i in range(want.pop("_n_dtrajs"))]
    dbg = compute_analysis_debug({"dtrajs": dtrajs}, lag=want.pop("_lag"), count_mode=want.pop("_mode"))
    np.testing.assert_array_equal(dbg.counts, g[f"{case}_counts"])
    _same(json.loads(json.dumps(dbg.to_summary_dict(), default=float)), want)


def test_dwell_runs_on_long_and_degenerate_sequences(engine):
    rng = np.random.default_rng(4)
    seq = np.repeat(rng.integers(0, 5, size=3000), rng.integers(1, 400, size=3000)).astype(np.int32)
    seq[100_000:100_050] = -1
    stats, rs, rl = engine.run_lengths(engine.to_device(seq), 5)
    edges = np.flatnonzero(np.diff(np.concatenate([[-7], seq, [-7]])) != 0)
    want = [(int(seq[a]), int(b - a)) for a, b in zip(edges[:-1], edges[1:]) if seq[a] >= 0]
    assert sorted(zip(rs.tolist(), rl.tolist())) == sorted(want)
    for s in range(5):
        lens = np.asarray([ln for st, ln in want if st == s])
        assert stats[:, s].tolist() == [lens.min(), lens.max(), lens.sum(), lens.size]
    stats, rs, rl = engine.run_lengths(engine.to_device(np.full(10, -1, np.int32)), 3)
    assert rs.size == 0 and stats[3].tolist() == [0, 0, 0] and stats[0].tolist() == [-1, -1, -1]


def test_debug_errors_and_scc():
    with pytest.raises(ValueError, match="no discrete trajectories"):
        compute_analysis_debug({"dtrajs": []}, lag=1)
    with pytest.raises(ValueError, match="no valid states"):
        compute_analysis_debug({"dtrajs": [np.full(10, -1)]}, lag=1)
    x = np.tile(np.arange(4), 500)
    x[::7] = -1                                  # many unassigned frames: the reference's own pair-count check trips
    with pytest.raises(ValueError, match="Pair counting mismatch"):
        compute_analysis_debug({"dtrajs": [x]}, lag=5)
    s = analyse_scc(np.array([[1, 1, 0], [1, 1, 0], [0, 1, 1.0]]))
    assert s.component_sizes.tolist() in ([2, 1], [1, 2]) and s.largest_component.tolist() == [0, 1]
    assert s.largest_fraction == pytest.approx(2 / 3)
