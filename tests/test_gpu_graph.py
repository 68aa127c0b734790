"""hipGraph capture of the hot path (msm_graph_begin / _end / _launch): the whole shard step and the
lag scan replay from a graph and reproduce the eager results bit for bit.  (north star: launch-bound
inner loops -- the 10 Lloyd iterations, the 50-lag scan -- go into hipGraphs.)"""
import time

import numpy as np
import pytest

from oracle import cport
from pmarlo_amd.dist import ShardConfig, ShardedMSM
from tests import _gen

pytestmark = pytest.mark.gpu


def test_step_replays_from_a_graph(engine):
    n, F, d, k, lag = 60_000, 32, 4, 100, 10
    cfg = ShardConfig(n_frames=n, n_features=F, tica_dim=d, k=k, lag=lag, kmeans_iters=6, seed=1)
    X = _gen.correlated_series(n, F, seed=5)
    msm = ShardedMSM(engine, cfg, engine.to_device(X))
    msm.step()                                   # eager: sizes every scratch buffer
    engine.sync()
    want_counts = msm.buf["counts"].to_host().copy()
    want_T = msm.T.to_host().copy()
    want_labels = msm.labels.to_host().copy()
    engine.graph_begin()
    msm.step()
    g = engine.graph_end()
    try:
        for _ in range(3):
            msm.buf["counts"].zero_()
            msm.T.zero_()
            msm.labels.zero_()
            engine.graph_launch(g)
            engine.sync()
            np.testing.assert_array_equal(msm.buf["counts"].to_host(), want_counts)
            np.testing.assert_array_equal(msm.T.to_host(), want_T)
            np.testing.assert_array_equal(msm.labels.to_host(), want_labels)
        # timing (informational): replay vs eager enqueue of the same ~60 launches
        t0 = time.perf_counter()
        for _ in range(20):
            engine.graph_launch(g)
        engine.sync()
        t_graph = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            msm.step()
        engine.sync()
        t_eager = (time.perf_counter() - t0) / 20
        print(f"step at n={n}: graph {t_graph * 1e3:.3f} ms, eager {t_eager * 1e3:.3f} ms")
    finally:
        engine.graph_destroy(g)


def test_lag_scan_replays_from_a_graph(engine):
    n, k = 40_000, 200
    lab = _gen.markov_labels(n, k, seed=2).astype(np.int32)
    ld = engine.to_device(lab)
    lags = list(range(1, 51))
    counts, pairs = engine.count_transitions_lagscan(ld, k, lags)            # eager once
    engine.sync()
    want = counts.to_host().copy()
    engine.graph_begin()
    engine.count_transitions_lagscan(ld, k, lags, out=counts, pairs=pairs)
    g = engine.graph_end()
    try:
        counts.zero_()
        engine.graph_launch(g)
        engine.sync()
        got = counts.to_host()
        np.testing.assert_array_equal(got, want)
        for i in (0, 9, 49):
            np.testing.assert_array_equal(got[i], cport.count_transitions(lab, k, lags[i])[0])
    finally:
        engine.graph_destroy(g)


def test_capture_refuses_growing_scratch():
    """Scratch growth needs a synchronising reallocation: refused under capture with a clear status,
    BEFORE any HIP call -- the capture stays valid and ends normally.  (Private engine; every buffer
    is allocated before the capture starts: hipMalloc is not a capturable operation.)"""
    import ctypes as C

    from pmarlo_amd import _lib
    from pmarlo_amd.device import Engine

    eng = Engine(0)
    try:
        n = 2500
        T = eng.to_device(np.full((n, n), 1.0 / n))
        Tk = eng.to_device(np.full((1, n, n), 1.0 / n))
        mse = eng.empty((1,), np.float64)
        fac = np.asarray([3], dtype=np.int32)
        lib = _lib.load()
        eng.sync()
        eng.graph_begin()
        st = lib.msm_ck_test(eng.handle, T.ptr, n, Tk.ptr, n * n, n, n, fac.ctypes.data, 1, None, n, mse.ptr, None)
        assert st == _lib.MSM_ERR_UNSUPPORTED        # 100 MB of matrix-power scratch would have to be allocated
        assert b"graph capture" in lib.msm_last_error(eng.handle)
        g = eng.graph_end()                          # the (empty) capture is still valid
        eng.graph_destroy(g)
        # and the same call works eagerly afterwards
        m, _ = eng.ck_test(T, Tk, [3])
        np.testing.assert_allclose(m, 0.0, atol=1e-30)
    finally:
        eng.close()
