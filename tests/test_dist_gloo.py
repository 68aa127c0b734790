"""N > 1 path on CPU: two gloo ranks drive the REAL pmarlo_amd.dist.ShardedMSM.step() -- the code bench.py runs for
N > 1 -- with a host stand-in for the engine (tests/_host_engine.py: numpy + the CPU oracle behind the Engine
interface) and TorchComm over gloo.  What can only go wrong with more than one rank is what is checked:
the shared shift broadcast, the single moments collective over the aliased buffers, the rank-ordered fp64 sum,
the one MIN collective that carries rank 0's centres and the coarsest fixed-point scale (and its reciprocal), the
out-of-place int64 member-sum reduction ordered against kmeans_update, the count reduction (single lag and the
batched lag scan)."""

from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]

N, F, D, K, LAG, ITERS = 6000, 8, 3, 12, 5, 4
LAGS = (1, 2, 5, 9)


def _cfg(rank: int, world: int, mode: str):
    from pmarlo_amd.dist import ShardConfig

    if mode == "tica":
        return ShardConfig(n_frames=N, n_features=F, tica_dim=D, k=K, lag=LAG, kmeans_iters=ITERS, seed=3, n_total=N * world)
    # C4-like: no TICA (cluster in the feature space), lag scan in one collective
    return ShardConfig(n_frames=N, n_features=F, tica_dim=0, k=K, lag=LAG, kmeans_iters=ITERS, seed=3, n_total=N * world,
                       lags=LAGS)


def _worker(rank: int, world: int, port: int, out_dir: str, mode: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist

    from pmarlo_amd.dist import ShardedMSM, TorchComm, exchange_aliases, exchange_shapes
    from tests import _gen
    from tests._host_engine import HostArray, HostEngine

    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _cfg(rank, world, mode)
    # exchange buffers: numpy memory shared by the torch tensors (collectives) and the engine's arrays
    arrays = {nm: np.zeros(shape, np.dtype(dt)) for nm, (shape, dt) in exchange_shapes(cfg).items()}
    tensors = {nm: torch.from_numpy(a) for nm, a in arrays.items()}
    views = {nm: HostArray(a) for nm, a in arrays.items()}
    for name, (parent, first, length) in exchange_aliases(cfg).items():
        tensors[name] = tensors[parent][first:first + length]
        views[name] = views[parent].view((length,), offset_elems=first)
    comm = TorchComm(tensors, views)
    eng = HostEngine()
    X = _gen.correlated_series(N, F, seed=1000 + rank)
    msm = ShardedMSM(eng, cfg, eng.to_device(X), comm=comm, shared=views)
    assert msm.collectives_per_step == (3 if mode == "tica" else 2) + ITERS
    before = comm.n_collectives
    msm.step()
    assert comm.n_collectives - before == msm.collectives_per_step
    msm.step()          # a second step must start from clean accumulators
    out = {nm: a.copy() for nm, a in arrays.items()}
    out["centers"] = arrays["start"][:K * cfg.cluster_dim].reshape(K, cfg.cluster_dim).copy()
    out.update(Y=np.asarray(msm.Y.a, np.float64).copy(), labels=msm.labels.a.copy(), T=msm.T.a.copy())
    if mode == "tica":
        out.update(eig=msm.eig.a.copy(), mean=msm.mean.a.copy(), scale=msm.scale.a.copy())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, mode):
    import torch.multiprocessing as mp

    world, port = 2, 29500 + (os.getpid() % 2000) + (0 if mode == "tica" else 1)
    mp.start_processes(_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True, start_method="spawn")
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]


def _replay_lloyd(Ys, centers0, scale, iters):
    """Lloyd iterations over ALL frames with the engine's fixed-point rule, written independently of dist.py."""
    from oracle import cport

    c = centers0.copy()
    k, d = c.shape
    for _ in range(iters):
        S = np.zeros((k, d), np.int64)
        cnt = np.zeros(k, np.int64)
        for Y in Ys:
            lab = cport.kmeans_assign(Y, c)
            np.add.at(S, lab, np.rint(Y * scale).astype(np.int64))
            cnt += np.bincount(lab, minlength=k)
        for j in range(k):
            if cnt[j] > 0:
                c[j] = S[j].astype(np.float64) * (1.0 / scale) / float(cnt[j])
    return c


def test_two_rank_step_with_tica(tmp_path):
    from oracle import cport, npport
    from tests import _gen
    from tests._host_engine import HostEngine, _splitmix_u

    g = _run(tmp_path, "tica")
    world = 2
    Xs = [_gen.correlated_series(N, F, seed=1000 + r).astype(np.float64) for r in range(world)]
    Xall = np.vstack(Xs)
    # every replicated quantity is bit-identical on the two ranks
    for key in ("shift", "moments", "centers", "counts", "eig", "T", "mean", "scale"):
        np.testing.assert_array_equal(g[0][key], g[1][key], err_msg=key)
    # fit state: slot 2 is the LOCAL max |Y| (never exchanged); scale, 2^-e, shift^2, tol, done, n_iter agree
    np.testing.assert_array_equal(np.delete(g[0]["fit_state"], 2), np.delete(g[1]["fit_state"], 2))
    # the shared shift is row 0 of rank 0's shard; standardisation = statistics of the concatenated data
    np.testing.assert_array_equal(g[0]["shift"], Xs[0][0])
    np.testing.assert_allclose(g[0]["mean"], Xall.mean(0), rtol=1e-12)
    np.testing.assert_allclose(g[0]["scale"], Xall.std(0), rtol=1e-12)
    # TICA on the list of shards (pairs never cross a shard) against the oracle's own formulation
    Xp = npport.preprocess(Xall, scale=True)
    model = npport.tica_fit([Xp[r * N:(r + 1) * N] for r in range(world)], LAG, dim=D)
    np.testing.assert_allclose(g[0]["eig"][:model["rank"]], model["eigenvalues"], rtol=1e-9, atol=1e-12)
    lag_block = g[0]["moments"][:2 * F * F + 2 * F + 1]
    assert lag_block[-1] == world * (N - LAG)
    # k-means: the scale every rank used is the coarsest one; the start is rank 0's stratified draw, on all ranks
    Ys = [g[r]["Y"] for r in range(world)]
    amax = [np.abs(Y).max() for Y in Ys]
    want_scale = min(np.ldexp(1.0, int(np.clip(61 - int(np.ceil(np.log2(N * world * a))), -900, 60))) for a in amax)
    assert g[0]["fit_state"][0] == want_scale and g[0]["fit_state"][0] * g[0]["fit_state"][1] == 1.0
    init = np.stack([Ys[0][min(int((j + _splitmix_u(3, j)) * (N / K)), N - 1)] for j in range(K)])
    np.testing.assert_array_equal(g[0]["centers"], _replay_lloyd(Ys, init, want_scale, ITERS))
    # labels are the assignment to the final centres; counts are the sum of the per-shard counts, exactly
    want = np.zeros((K, K), np.int64)
    for r in range(world):
        np.testing.assert_array_equal(g[r]["labels"], cport.kmeans_assign(Ys[r], g[0]["centers"]))
        want += cport.count_transitions(g[r]["labels"], K, LAG)[0]
    np.testing.assert_array_equal(g[0]["counts"][:K * K].reshape(K, K), want)
    assert int(g[0]["counts"][K * K]) == world * (N - LAG)
    np.testing.assert_array_equal(g[0]["T"], npport.normalise_counts(want.astype(np.float64)))
    # the exchange buffer holds the all-reduced member sums of the last Lloyd pass (the sums are incremental and
    # persist over the passes; every step zeroes them first -- the worker ran two steps and the centres above are
    # those of a clean run): every frame is booked under exactly one centre
    assert int(g[0]["km_acc"][K * D:].sum()) == world * N and (g[0]["km_acc"][K * D:] >= 0).all()


def test_two_rank_step_without_tica_lag_scan(tmp_path):
    """BASELINE config 4 shape: clustering in the feature space, ITS lag scan -> ONE L x k x k int64 collective."""
    from oracle import cport, npport

    g = _run(tmp_path, "lagscan")
    world, L = 2, len(LAGS)
    for key in ("centers", "counts", "T"):
        np.testing.assert_array_equal(g[0][key], g[1][key], err_msg=key)
    np.testing.assert_array_equal(np.delete(g[0]["fit_state"], 2), np.delete(g[1]["fit_state"], 2))
    cnt = g[0]["counts"]
    for i, lag in enumerate(LAGS):
        want = sum(cport.count_transitions(g[r]["labels"], K, lag)[0] for r in range(world))
        np.testing.assert_array_equal(cnt[i * K * K:(i + 1) * K * K].reshape(K, K), want)
        assert int(cnt[L * K * K + i]) == world * (N - lag)
    li = LAGS.index(LAG)
    np.testing.assert_array_equal(g[0]["T"], npport.normalise_counts(
        cnt[li * K * K:(li + 1) * K * K].reshape(K, K).astype(np.float64)))
    for r in range(world):
        np.testing.assert_array_equal(g[r]["labels"], cport.kmeans_assign(g[r]["Y"], g[0]["centers"]))


def test_exchange_payload_sizes_match_survey():
    """C3 payloads (F=64, d=10, k=500) and the C4 lag scan (k=200, L=50: 16 MB)."""
    from pmarlo_amd.dist import ShardConfig, exchange_shapes

    sh = exchange_shapes(ShardConfig(n_frames=1_000_000, n_features=64, tica_dim=10, k=500, lag=10))
    nbytes = {k: int(np.prod(s)) * 8 for k, (s, _) in sh.items()}
    assert nbytes["moments"] == (2 * 64 * 64 + 2 * 64 + 1 + 3 * 64) * 8   # ~68 KB: lagged moments + the sums
    assert nbytes["km_acc"] == (500 * 10 + 500) * 8             # 44 KB per Lloyd iteration
    assert nbytes["start"] == (500 * 10 + 1) * 8                # centres + scale: one MIN collective
    assert nbytes["counts"] == (500 * 500 + 1) * 8              # 2 MB (+ the pair count)
    sh4 = exchange_shapes(ShardConfig(n_frames=100_000, n_features=45, tica_dim=0, k=200, lag=1, lags=tuple(range(1, 51))))
    assert int(np.prod(sh4["counts"][0])) * 8 == (50 * 200 * 200 + 50) * 8   # one 16 MB collective
    assert sh4["start"][0] == (200 * 45 + 1,)
