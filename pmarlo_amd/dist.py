"""One trajectory shard per GPU: the hot path with all-reduce of the SMALL buffers only.

Shards are independent trajectory segments (lag pairs never cross a shard),
so the big arrays (features, projected coordinates, labels) never leave their GPU.  What is
exchanged, by plain summation over RCCL/xGMI (``torch.distributed``, backend "nccl"):

  exchange                      payload                         when
  lagged moments + the          2F^2 + 2F + 1 + 3F f64          once (one buffer, one collective)
    standardisation sums
  k-means fixed-point scale     1 f64 (MIN) + centres bcast     once
  k-means member sums / counts  k*d + k int64 (exact)           per Lloyd iteration
  transition counts             k^2 int64 (exact)               once

Integer payloads make the result independent of the number of shards bit for bit; the fp64
moment sums are gathered and added in rank order (bit-identical on every rank, independent of the
collective's schedule).  With ``comm=None`` the
same code runs on a single GPU (bench.py at N=1, tests).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .device import DeviceArray, Engine

__all__ = ["Comm", "ShardedMSM", "ShardConfig", "exchange_shapes", "exchange_aliases", "TorchComm",
           "torch_exchange_buffers"]


class Comm:
    """Reduction interface over device buffers.  ``TorchComm`` wraps torch.distributed; the
    buffers it is given are views of torch tensors (see ``ShardedMSM.alloc``)."""

    world = 1
    rank = 0

    def allreduce_sum(self, name: str) -> None: ...
    def allreduce_min(self, name: str) -> None: ...
    def allreduce_max(self, name: str) -> None: ...
    def broadcast(self, name: str, src: int = 0) -> None: ...
    def reciprocal(self, dst: str, src: str) -> None: ...


class TorchComm(Comm):
    """torch.distributed collectives on named torch tensors (device or, for gloo tests, host)."""

    def __init__(self, tensors: dict):
        import torch.distributed as dist

        self.dist = dist
        self.t = tensors
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self._gather: dict = {}
        self._into_tensor = dist.get_backend() == "nccl"

    def allreduce_sum(self, name):
        t = self.t[name]
        if not t.dtype.is_floating_point:
            # integer sums commute: any reduction schedule gives the same bits
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            return
        # fp64 moment sums: gather every rank's buffer and add them in rank order, so the result
        # does not depend on the collective's internal schedule and is bit-identical on all ranks
        # (these buffers are <= a few hundred KB: latency-bound either way)
        import torch

        buf = self._gather.get(name)
        if buf is None:
            buf = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            self._gather[name] = buf
        # the collective form was fixed at construction from the backend name; an exception here is a
        # real communicator / device failure and propagates (the rank exits non-zero)
        if self._into_tensor:
            self.dist.all_gather_into_tensor(buf, t)            # one RCCL call, no per-rank tensor list
        else:
            self.dist.all_gather([buf[r] for r in range(self.world)], t)
        # fixed-order sum over the rank axis: one kernel, no atomics, the same bits on every rank
        torch.sum(buf, dim=0, out=t)

    def allreduce_min(self, name):
        self.dist.all_reduce(self.t[name], op=self.dist.ReduceOp.MIN)

    def allreduce_max(self, name):
        self.dist.all_reduce(self.t[name], op=self.dist.ReduceOp.MAX)

    def broadcast(self, name, src=0):
        self.dist.broadcast(self.t[name], src=src)

    def reciprocal(self, dst, src):
        """t[dst] = 1 / t[src] on the exchange stream (a torch elementwise op, ordered with the collectives)."""
        import torch

        torch.reciprocal(self.t[src], out=self.t[dst])


@dataclass
class ShardConfig:
    n_frames: int        # frames of THIS shard
    n_features: int
    tica_dim: int
    k: int
    lag: int
    kmeans_iters: int = 10
    seed: int = 0
    n_total: int | None = None   # frames over all shards (defaults to n_frames * world)


def exchange_shapes(cfg: ShardConfig) -> dict[str, tuple[tuple[int, ...], str]]:
    """name -> (shape, dtype) of every buffer that crosses the interconnect."""
    F, d, k = cfg.n_features, cfg.tica_dim, cfg.k
    return {
        "shift": ((F,), "float64"),
        # [lagged moments 2F^2 + 2F + 1 | standardisation sums 3F]: both are raw sums about the shared shift
        "moments": ((2 * F * F + 2 * F + 1 + 3 * F,), "float64"),
        "fit_state": ((8,), "float64"),
        "centers": ((k, d), "float64"),
        "km_acc": ((k * d + k,), "int64"),
        "counts": ((k * k + 1,), "int64"),   # the pair count rides at the end: one collective
    }


class ShardedMSM:
    """featurised shard -> TICA -> k-means -> counts -> T, device resident, one step = one pass."""

    def __init__(self, engine: Engine, cfg: ShardConfig, x: DeviceArray, comm: Comm | None = None,
                 shared: dict[str, DeviceArray] | None = None, always_exchange: bool = False):
        self.eng, self.cfg, self.x = engine, cfg, x
        self.comm = comm
        # always_exchange: run every collective even in a group of one (rehearses the RCCL calls on a
        # one-GPU box; a sum / min / broadcast over one rank leaves the buffers bit-identical)
        self.always_exchange = bool(always_exchange) and comm is not None
        self.world = comm.world if comm else 1
        self.n_total = cfg.n_total if cfg.n_total is not None else cfg.n_frames * self.world
        eng = engine
        F, d, k, n = cfg.n_features, cfg.tica_dim, cfg.k, cfg.n_frames
        self.buf = shared if shared is not None else {
            nm: eng.zeros(shape, np.dtype(dt)) for nm, (shape, dt) in exchange_shapes(cfg).items()}
        b = self.buf
        for name, (parent, first, length) in exchange_aliases(cfg).items():
            if name not in b:
                b[name] = b[parent].view((length,), offset_elems=first)
        self.mean, self.scale, self.inv_scale = (eng.empty((F,), np.float64) for _ in range(3))
        self.eig = eng.empty((F,), np.float64)
        self.W = eng.empty((F, F), np.float64)
        self.m2 = eng.empty((F,), np.float64)
        self.rank_d = eng.empty((1,), np.int32)
        self.Y = eng.empty((n, d), np.float64)
        self.labels = eng.empty((n,), np.int32)
        self.T = eng.empty((k, k), np.float64)
        self.rowsum = eng.empty((k,), np.float64)
        self.diag = eng.empty((1,), np.float64)
        self.km_sums = b["km_acc"].view((k * d,), np.int64)
        self.km_counts = b["km_acc"].view((k,), np.int64, offset_elems=k * d)
        # bf16 frame images of the k-means filter: built once per step after the projection, read by all Lloyd
        # passes and the final assignment (None when d / k are outside the filter's range)
        nbytes = eng.kmeans_image_bytes(n, d)
        self.km_image = eng.empty((nbytes,), np.uint8) if nbytes else None
        self.accum_events: list = []
        self.time_accum = False
        # one shift vector shared by all shards (row 0 of rank 0's shard) so that the raw
        # moment sums add across ranks
        _, first = eng.column_moments_partial(x)
        check_d2d = first.to_host()
        b["shift"].copy_from_host(check_d2d)
        if comm and (comm.world > 1 or self.always_exchange):
            comm.broadcast("shift", 0)

    def step(self) -> None:
        from ._lib import check, lib

        eng, cfg, b, comm = self.eng, self.cfg, self.buf, self.comm
        multi = comm is not None and (comm.world > 1 or self.always_exchange)
        F, d, k = cfg.n_features, cfg.tica_dim, cfg.k
        # 1. time-lagged raw moments about the shared shift (fp64 MFMA): the only pass over X before the
        #    projection -- the standardisation sums follow from them and 2 * lag edge frames
        eng.lagged_moments(self.x, cfg.lag, b["shift"], assume_finite=True, out=b["lagged"])
        eng.moments_from_lagged(self.x, cfg.lag, b["shift"], b["lagged"], out=b["mom_sums"])
        if multi:
            comm.allreduce_sum("moments")
        eng.standardise_params(b["mom_sums"], b["shift"], F, float(self.n_total), True,
                               out=(self.mean, self.scale, self.inv_scale))
        # 2. TICA solve
        check(lib.msm_tica_solve(eng.handle, b["lagged"].ptr, self.scale.ptr, F, 1e-6, 1, self.eig.ptr, self.W.ptr,
                                 self.m2.ptr, self.rank_d.ptr), eng.handle)
        # 3. projection
        # max |Y| (the fixed-point scale of the Lloyd sums needs it) falls out of the same pass
        # (x - shift) / sigma - m2: m2 is the symmetric mean about the SAME shift the moments used
        eng.project(self.x, b["shift"], self.inv_scale, self.W, d, mean2=self.m2, out=self.Y,
                    absmax=b["fit_state"].view((1,), offset_elems=2))
        # 4. k-means: fixed number of Lloyd iterations over all frames
        check(lib.msm_kmeans_fit_begin(eng.handle, self.Y.ptr, 1, cfg.n_frames, d, d, None, None, k, cfg.seed, 1,
                                       float(self.n_total), 0.0, b["centers"].ptr, b["fit_state"].ptr, 1), eng.handle)
        if multi:
            # identical start on every rank: rank 0's centres; the coarsest fixed-point scale
            # (state = {scale, inv_scale, ...}: MIN of scale, inv_scale follows as MAX)
            comm.broadcast("centers", 0)
            comm.allreduce_min("fit_scale")
            comm.reciprocal("fit_inv_scale", "fit_scale")   # 2^-e: exact, no second collective
        check(lib.msm_memset(eng.handle, b["km_acc"].ptr, 0, b["km_acc"].nbytes), eng.handle)
        if self.km_image is not None:
            eng.kmeans_pack(self.Y, image=self.km_image)
        for _ in range(cfg.kmeans_iters):
            if self.time_accum:
                e0, e1 = eng.event(), eng.event()
                e0.record()
            eng.kmeans_accumulate(self.Y, b["centers"], b["fit_state"], self.km_sums, self.km_counts,
                                  image=self.km_image)
            if self.time_accum:
                e1.record()
                self.accum_events.append((e0, e1))
            if multi:
                comm.allreduce_sum("km_acc")
            eng.kmeans_update(self.km_sums, self.km_counts, b["centers"], b["fit_state"], clear=True)
        eng.kmeans_assign(self.Y, b["centers"], labels=self.labels, image=self.km_image)
        # 5. lag-tau counts + row-normalised transition matrix
        eng.count_transitions(self.labels, k, cfg.lag, out=b["counts"].view((k, k)),
                              pairs=b["counts"].view((1,), offset_elems=k * k))
        if multi:
            comm.allreduce_sum("counts")
        check(lib.msm_transition_matrix(eng.handle, b["counts"].ptr, 0, k, 0, 0.0, 0.0, self.T.ptr, None, None, None,
                                        self.rowsum.ptr, self.diag.ptr), eng.handle)


def exchange_aliases(cfg: ShardConfig) -> dict[str, tuple[str, int, int]]:
    """name -> (parent buffer, first element, length) of the named parts of exchange buffers."""
    F = cfg.n_features
    L = 2 * F * F + 2 * F + 1
    return {"lagged": ("moments", 0, L), "mom_sums": ("moments", L, 3 * F),
            "fit_scale": ("fit_state", 0, 1), "fit_inv_scale": ("fit_state", 1, 1)}


def torch_exchange_buffers(engine: Engine, cfg: ShardConfig, device) -> tuple[dict, dict]:
    """Allocate the exchange buffers as torch tensors on `device` (so torch.distributed can
    reduce them) and return (tensors, DeviceArray views for the engine).  The entries of
    exchange_aliases() ("lagged", "mom_sums", "fit_scale", "fit_inv_scale") alias parts of their parents."""
    import torch

    tensors, views = {}, {}
    for name, (shape, dt) in exchange_shapes(cfg).items():
        t = torch.zeros(shape, dtype=getattr(torch, dt), device=device)
        tensors[name] = t
        views[name] = engine.wrap(t.data_ptr(), shape, np.dtype(dt))
    for name, (parent, first, length) in exchange_aliases(cfg).items():
        tensors[name] = tensors[parent][first:first + length]
        views[name] = views[parent].view((length,), offset_elems=first)
    return tensors, views
