"""N > 1 path on CPU: world_size-2 gloo run of the exchange protocol of pmarlo_amd.dist.

The HIP kernels cannot run here, so each rank computes its shard's partial statistics with the
CPU oracle (stand-in for the engine), pushes them through the SAME TorchComm collectives and
exchange buffers the GPU path uses, and the merged result must equal the single-shard result:
  - moment sums / lagged moments: equal to summation order (fp64)
  - k-means fixed-point sums / counts and transition counts: bit-exact (int64)
This is what makes the result independent of the number of GPUs."""

from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist

    from oracle import cport, npport
    from pmarlo_amd.dist import ShardConfig, TorchComm, exchange_aliases, exchange_shapes
    from tests import _gen

    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, F, d, k, lag = 6000, 8, 3, 12, 5
    cfg = ShardConfig(n_frames=n, n_features=F, tica_dim=d, k=k, lag=lag, n_total=n * world)
    tensors = {nm: torch.zeros(shape, dtype=getattr(torch, dt)) for nm, (shape, dt) in exchange_shapes(cfg).items()}
    for name, (parent, first, length) in exchange_aliases(cfg).items():
        tensors[name] = tensors[parent][first:first + length]
    comm = TorchComm(tensors)
    assert comm.world == world and comm.rank == rank

    X = _gen.correlated_series(n, F, seed=1000 + rank).astype(np.float64)
    # shared shift = row 0 of rank 0
    if rank == 0:
        tensors["shift"].copy_(torch.from_numpy(X[0].copy()))
    comm.broadcast("shift", 0)
    shift = tensors["shift"].numpy().copy()
    dlt = X - shift
    # the step's protocol: lagged moments and standardisation sums about the shared shift, ONE collective
    tensors["mom_sums"].copy_(torch.from_numpy(np.concatenate([np.full(F, float(n)), dlt.sum(0), (dlt ** 2).sum(0)])))
    m = npport.lagged_moments([dlt], lag)
    tensors["lagged"].copy_(torch.from_numpy(np.concatenate([m["Mxx"].ravel(), m["Mxy_half"].ravel(), m["sx"], m["sy"],
                                                              [float(m["T"])]])))
    comm.allreduce_sum("moments")
    s = tensors["mom_sums"].numpy()
    cnt, s1, s2 = s[:F], s[F:2 * F], s[2 * F:]
    mean = shift + s1 / cnt
    sigma = np.sqrt((s2 - s1 * s1 / cnt) / (n * world))
    # k-means exchange: scale MIN, centres broadcast, int64 sums
    Y = (X - mean) / sigma
    Y = Y[:, :d].copy()
    amax = np.abs(Y).max()
    e = 61 - int(np.ceil(np.log2(n * world * amax)))
    tensors["fit_state"][0] = float(np.ldexp(1.0, e))
    tensors["fit_state"][1] = float(np.ldexp(1.0, -e))
    comm.allreduce_min("fit_scale")
    comm.reciprocal("fit_inv_scale", "fit_scale")
    scale = float(tensors["fit_state"][0])
    assert float(tensors["fit_state"][1]) * scale == 1.0   # power of two: the reciprocal is exact
    if rank == 0:
        tensors["centers"].copy_(torch.from_numpy(Y[:: n // k][:k].copy()))
    comm.broadcast("centers", 0)
    centers = tensors["centers"].numpy().copy()
    lab = cport.kmeans_assign(Y, centers)
    acc = np.zeros(k * d + k, np.int64)
    np.add.at(acc[:k * d].reshape(k, d), lab, np.rint(Y * scale).astype(np.int64))
    acc[k * d:] = np.bincount(lab, minlength=k)
    tensors["km_acc"].copy_(torch.from_numpy(acc))
    comm.allreduce_sum("km_acc")
    c, p = cport.count_transitions(lab, k, lag)
    tensors["counts"][:k * k].copy_(torch.from_numpy(c.ravel()))
    tensors["counts"][k * k] = p        # the pair count rides at the end of the counts buffer
    comm.allreduce_sum("counts")
    if rank == 0:
        np.savez(os.path.join(out_dir, "merged.npz"), **{k_: v.numpy() for k_, v in tensors.items()}, mean=mean,
                 sigma=sigma, scale=scale)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_equals_single_shard(tmp_path):
    import torch.multiprocessing as mp

    from oracle import cport, npport
    from tests import _gen

    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    g = np.load(tmp_path / "merged.npz")
    n, F, d, k, lag = 6000, 8, 3, 12, 5
    Xs = [_gen.correlated_series(n, F, seed=1000 + r).astype(np.float64) for r in range(world)]
    Xall = np.vstack(Xs)
    # standardisation: equals the statistics of the concatenated data
    np.testing.assert_allclose(g["mean"], Xall.mean(0), rtol=1e-12)
    np.testing.assert_allclose(g["sigma"], Xall.std(0), rtol=1e-12)
    # lagged moments: per-shard pairs only (no pair crosses the shard boundary)
    ref = npport.lagged_moments([X - g["shift"] for X in Xs], lag)
    lagged = g["lagged"]
    np.testing.assert_allclose(lagged[:F * F].reshape(F, F), ref["Mxx"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(lagged[F * F:2 * F * F].reshape(F, F), ref["Mxy_half"], rtol=1e-12, atol=1e-9)
    assert lagged[-1] == ref["T"] == world * (n - lag)
    # integer payloads: bit-exact, independent of the number of shards
    Y = ((Xall - g["mean"]) / g["sigma"])[:, :d].copy()
    lab = cport.kmeans_assign(Y, g["centers"])
    acc = np.zeros(k * d + k, np.int64)
    np.add.at(acc[:k * d].reshape(k, d), lab, np.rint(Y * float(g["scale"])).astype(np.int64))
    acc[k * d:] = np.bincount(lab, minlength=k)
    np.testing.assert_array_equal(g["km_acc"], acc)
    want = sum(cport.count_transitions(lab[r * n:(r + 1) * n], k, lag)[0] for r in range(world))
    np.testing.assert_array_equal(g["counts"][:k * k].reshape(k, k), want)
    assert int(g["counts"][k * k]) == world * (n - lag)
    # scale agreed on by all ranks is the coarsest one
    amax = max(np.abs(((X - g["mean"]) / g["sigma"])[:, :d]).max() for X in Xs)
    assert float(g["scale"]) <= np.ldexp(1.0, 61 - int(np.ceil(np.log2(n * world * amax)))) * (1 + 1e-15)


def test_exchange_payload_sizes_match_survey():
    """C3 payloads (F=64, d=10, k=500)."""
    from pmarlo_amd.dist import ShardConfig, exchange_shapes

    sh = exchange_shapes(ShardConfig(n_frames=1_000_000, n_features=64, tica_dim=10, k=500, lag=10))
    nbytes = {k: int(np.prod(s)) * 8 for k, (s, _) in sh.items()}
    assert nbytes["moments"] == (2 * 64 * 64 + 2 * 64 + 1 + 3 * 64) * 8   # ~68 KB: lagged moments + the sums
    assert nbytes["km_acc"] == (500 * 10 + 500) * 8             # 44 KB per Lloyd iteration
    assert nbytes["counts"] == (500 * 500 + 1) * 8              # 2 MB (+ the pair count)
