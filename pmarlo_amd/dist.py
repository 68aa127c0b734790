"""One trajectory shard per GPU: the hot path with all-reduce of the SMALL buffers only.

Shards are independent trajectory segments (lag pairs never cross a shard),
so the big arrays (coordinates, features, projected coordinates, labels) never leave their GPU.
What is exchanged, by plain summation over RCCL / xGMI:

  exchange                      payload                         when
  lagged moments + the          2F^2 + 2F + 1 + 3F f64          once (one buffer, one collective)
    standardisation sums
  k-means start                 k*d + 1 f64, ONE MIN: rank 0's  once
                                centres (the other ranks offer
                                1.4e306) and the coarsest
                                fixed-point scale
  k-means member sums / counts  k*d + k int64 (exact), reduced  per Lloyd iteration
                                out of the persistent local sums
  transition counts             k^2 int64 (exact), or the       once (one collective for a whole
                                L k^2 + L block of a lag scan     lag scan)

i.e. 3 + kmeans_iters collectives per step with TICA, 2 + kmeans_iters without (`ShardedMSM.collectives_per_step`).
Integer payloads make the result independent of the number of shards bit for bit; the fp64 moment sums are
gathered and added in rank order (bit-identical on every rank, independent of the collective's schedule).

Two transports implement `Comm`:
  NativeComm   RCCL through the C ABI (msm_comm_init / msm_allreduce_* / msm_broadcast, include/msmhip.h) on the
               engine's stream; the 128-byte RCCL id travels through a file (`bootstrap_id`).  No torch involved.
  TorchComm    torch.distributed collectives on torch tensors (the gloo tests on CPU; "nccl" also works).
With ``comm=None`` the same code runs on a single GPU (bench.py at N = 1, tests).
"""

from __future__ import annotations

import ctypes as C
import os
import time
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from .device import DeviceArray, Engine

__all__ = ["Comm", "NativeComm", "TorchComm", "ShardedMSM", "ShardConfig", "exchange_shapes", "exchange_aliases",
           "torch_exchange_buffers", "bootstrap_id"]


class Comm:
    """Reduction interface over NAMED exchange buffers (exchange_shapes / exchange_aliases)."""

    world = 1
    rank = 0

    def allreduce_sum(self, name: str) -> None: ...
    def allreduce_sum_from(self, name: str, src) -> None: ...     # buffer `name` = sum over ranks of the array `src`
    def allreduce_min(self, name: str) -> None: ...
    def allreduce_max(self, name: str) -> None: ...
    def broadcast(self, name: str, src: int = 0) -> None: ...
    def reciprocal(self, dst: str, src: str) -> None: ...


def _id_file_default() -> Path:
    """Node-local file that carries the RCCL id of THIS launch: keyed by the rendezvous port, the launcher's process id
    and, under torchrun, the run id and restart count (a restarted group must not read the id of the group it replaces)."""
    env = os.environ
    key = "_".join([env.get("MASTER_PORT", "0"), str(os.getppid()), env.get("TORCHELASTIC_RUN_ID", "none"),
                    env.get("TORCHELASTIC_RESTART_COUNT", "0")])
    key = "".join(ch if ch.isalnum() or ch in "_-" else "-" for ch in key)
    return Path(env.get("MSM_COMM_ID_DIR", "/tmp")) / f"msm_comm_{key}.id"


def _resolve_id_path(world: int) -> Path:
    path = os.environ.get("MSM_COMM_ID_FILE")
    if path is not None:
        return Path(path)
    nnodes = int(os.environ.get("GROUP_WORLD_SIZE", os.environ.get("NNODES", "1")) or 1)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)) or world)
    if nnodes > 1 or local_world < world:
        raise RuntimeError("the RCCL id travels through a node-local file by default: set MSM_COMM_ID_FILE to a path "
                           "every node can read for a multi-node launch")
    return _id_file_default()


def bootstrap_id(rank: int, world: int, path: str | os.PathLike | None = None, timeout: float = 300.0) -> bytes:
    """Carry rank 0's RCCL unique id to the other ranks of this launch through a file.

    MSM_COMM_ID_FILE names the file (it must be visible to every rank: a shared file system when the job spans nodes);
    without it the file is node-local (`_id_file_default`), so a multi-node launch is refused rather than left to time
    out.  Rank 0 removes whatever an earlier launch left under the name before it draws the id, and the file is removed
    again once the communicator stands (`NativeComm.from_env`)."""
    from ._lib import check, lib

    path = Path(path) if path is not None else _resolve_id_path(world)
    if rank == 0:
        buf = (C.c_ubyte * 128)()
        check(lib.msm_comm_unique_id(buf, 128), None)
        tmp = path.with_suffix(f".tmp{os.getpid()}")
        tmp.write_bytes(bytes(buf))
        os.replace(tmp, path)          # atomic: a reader sees the old file, no file, or all 128 new bytes
        return bytes(buf)
    t0 = time.monotonic()
    started = time.time()
    while True:
        try:
            # a file older than this process belongs to an earlier launch that used the same name
            if path.stat().st_mtime >= started - float(os.environ.get("MSM_COMM_ID_MAX_AGE", "120")):
                data = path.read_bytes()
                if len(data) == 128:
                    return data
        except FileNotFoundError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no RCCL id at {path} after {timeout:.0f} s")
        time.sleep(0.01)


class NativeComm(Comm):
    """RCCL collectives through the C ABI, on the engine's stream, in place on DeviceArrays."""

    def __init__(self, engine: Engine, bufs: dict[str, DeviceArray], rank: int, world: int, uid: bytes,
                 id_file: str | os.PathLike | None = None):
        from ._lib import check, lib

        self.eng, self.b, self.rank, self.world = engine, bufs, int(rank), int(world)
        self._lib, self._check = lib, check
        handle = C.c_void_p()
        idbuf = (C.c_ubyte * 128).from_buffer_copy(uid)
        check(lib.msm_comm_init(engine.handle, self.rank, self.world, idbuf, 128, C.byref(handle)), engine.handle)
        self.handle = handle
        self._id_file = id_file
        # per-collective timing (bench.py): HIP events on the engine's stream around every call, keyed by buffer name
        self.timing = False
        self.events: dict[str, list] = {}

    @classmethod
    def from_env(cls, engine: Engine, bufs: dict[str, DeviceArray]) -> "NativeComm":
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        path = _resolve_id_path(world)
        if rank == 0:
            try:
                os.unlink(path)      # nothing of an earlier launch may sit under this name
            except OSError:
                pass
        self = cls(engine, bufs, rank, world, bootstrap_id(rank, world, path), id_file=path)
        # the communicator stands on every rank once a collective has completed: the id file has done its work
        if "timing" in bufs:
            self.allreduce_max("timing")
            engine.sync()
            if rank == 0:
                try:
                    os.unlink(path)
                except OSError:
                    pass
        return self

    def comm_info(self) -> tuple[int, int]:
        """(rank, world) as the communicator itself reports them."""
        r, w = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.msm_comm_info(self.handle, C.byref(r), C.byref(w), None), self.eng.handle)
        return int(r.value), int(w.value)

    def _timed(self, name, call):
        if not self.timing:
            return call()
        e0, e1 = self.eng.event(), self.eng.event()
        e0.record()
        call()
        e1.record()
        self.events.setdefault(name, []).append((e0, e1))

    def exchange_ms(self) -> dict[str, float]:
        """Device time between the events around the collectives of each named buffer, summed (call after a sync)."""
        return {nm: float(sum(a.elapsed_ms(b) for a, b in evs)) for nm, evs in self.events.items()}

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._lib.msm_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_collectives(self) -> int:
        n = C.c_uint64(0)
        self._check(self._lib.msm_comm_info(self.handle, None, None, C.byref(n)), self.eng.handle)
        return int(n.value)

    def allreduce_sum(self, name):
        a = self.b[name]
        fn = self._lib.msm_allreduce_i64 if a.dtype == np.int64 else self._lib.msm_allreduce_f64
        if a.dtype not in (np.dtype(np.int64), np.dtype(np.float64)):
            raise TypeError(f"exchange buffer {name!r} must be int64 or float64")
        self._timed(name, lambda: self._check(fn(self.handle, a.ptr, a.size), self.eng.handle))

    def allreduce_sum_from(self, name, src):
        a = self.b[name]
        if a.dtype != np.dtype(np.int64) or src.dtype != a.dtype or src.size != a.size:
            raise TypeError(f"exchange buffer {name!r}: the out-of-place sum takes int64 arrays of one size")
        self._timed(name, lambda: self._check(self._lib.msm_allreduce_i64_from(self.handle, src.ptr, a.ptr, a.size),
                                              self.eng.handle))

    def allreduce_min(self, name):
        a = self.b[name]
        self._timed(name, lambda: self._check(self._lib.msm_allreduce_min_f64(self.handle, a.ptr, a.size), self.eng.handle))

    def allreduce_max(self, name):
        a = self.b[name]
        self._timed(name, lambda: self._check(self._lib.msm_allreduce_max_f64(self.handle, a.ptr, a.size), self.eng.handle))

    def broadcast(self, name, src=0):
        a = self.b[name]
        self._timed(name, lambda: self._check(self._lib.msm_broadcast(self.handle, a.ptr, a.nbytes, int(src)),
                                              self.eng.handle))

    def reciprocal(self, dst, src):
        self.eng.rcp(self.b[src], self.b[dst])


class TorchComm(Comm):
    """torch.distributed collectives on named torch tensors (device or, for gloo tests, host)."""

    def __init__(self, tensors: dict, views: dict | None = None):
        """`views`: the engine's arrays over the same memory as `tensors` (needed for `allreduce_sum_from`)."""
        import torch.distributed as dist

        self.dist = dist
        self.t = tensors
        self.v = views
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self._gather: dict = {}
        self._into_tensor = dist.get_backend() == "nccl"
        self.n_collectives = 0

    def allreduce_sum(self, name):
        t = self.t[name]
        self.n_collectives += 1
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
        acc = buf[0].clone()
        for r in range(1, self.world):
            acc += buf[r]
        t.copy_(acc)

    def allreduce_sum_from(self, name, src):
        """t[name] = sum over ranks of the engine array `src`: torch.distributed reduces in place, so the engine copies
        `src` into the exchange buffer first (same stream as the collective)."""
        if self.v is None:
            raise RuntimeError("TorchComm needs the engine's views of the exchange buffers for an out-of-place sum")
        self.v[name].copy_from(src)
        self.n_collectives += 1
        self.dist.all_reduce(self.t[name], op=self.dist.ReduceOp.SUM)

    def allreduce_min(self, name):
        self.n_collectives += 1
        self.dist.all_reduce(self.t[name], op=self.dist.ReduceOp.MIN)

    def allreduce_max(self, name):
        self.n_collectives += 1
        self.dist.all_reduce(self.t[name], op=self.dist.ReduceOp.MAX)

    def broadcast(self, name, src=0):
        self.n_collectives += 1
        self.dist.broadcast(self.t[name], src=src)

    def reciprocal(self, dst, src):
        """t[dst] = 1 / t[src] on the exchange stream (a torch elementwise op, ordered with the collectives)."""
        import torch

        torch.reciprocal(self.t[src], out=self.t[dst])


@dataclass
class ShardConfig:
    n_frames: int        # frames of THIS shard
    n_features: int
    tica_dim: int        # 0: no TICA, cluster in the feature space itself (C4 / C5 of BASELINE.json)
    k: int
    lag: int
    kmeans_iters: int = 10
    seed: int = 0
    n_total: int | None = None   # frames over all shards (defaults to n_frames * world)
    lags: tuple | None = None    # lag scan: count matrices for all these lags in one pass, one collective
    n_atoms: int = 0             # > 0: the shard is xyz float32 [n, n_atoms, 3]; features = pair distances
    pairs: np.ndarray | None = None   # int32 [n_features, 2] atom pairs of the featurize front stage

    @property
    def cluster_dim(self) -> int:
        return self.tica_dim if self.tica_dim > 0 else self.n_features


def exchange_shapes(cfg: ShardConfig) -> dict[str, tuple[tuple[int, ...], str]]:
    """name -> (shape, dtype) of every buffer that crosses the interconnect."""
    F, d, k = cfg.n_features, cfg.cluster_dim, cfg.k
    L = len(cfg.lags) if cfg.lags else 1
    return {
        "shift": ((F,), "float64"),
        # [lagged moments 2F^2 + 2F + 1 | standardisation sums 3F]: both are raw sums about the shared shift
        "moments": ((2 * F * F + 2 * F + 1 + 3 * F,), "float64"),
        "fit_state": ((8,), "float64"),
        # [centres k d | fixed-point scale]: one MIN collective gives every rank rank 0's centres and the coarsest scale
        "start": ((k * d + 1,), "float64"),
        "km_acc": ((k * d + k,), "int64"),
        "counts": ((L * k * k + L,), "int64"),   # the pair counts ride at the end: one collective
    }


def exchange_aliases(cfg: ShardConfig) -> dict[str, tuple[str, int, int]]:
    """name -> (parent buffer, first element, length) of the named parts of exchange buffers."""
    F, d, k = cfg.n_features, cfg.cluster_dim, cfg.k
    L = 2 * F * F + 2 * F + 1
    return {"lagged": ("moments", 0, L), "mom_sums": ("moments", L, 3 * F),
            "fit_scale": ("fit_state", 0, 1), "fit_inv_scale": ("fit_state", 1, 1),
            "centers": ("start", 0, k * d), "start_scale": ("start", k * d, 1)}


class ShardedMSM:
    """(xyz ->) features -> (TICA ->) k-means -> counts -> T, device resident, one step = one pass.

    Only `Engine` methods are called (no direct library calls), so a host stand-in engine can drive this very
    code on CPU (tests/test_dist_gloo.py)."""

    def __init__(self, engine: Engine, cfg: ShardConfig, x: DeviceArray, comm: Comm | None = None,
                 shared: dict[str, DeviceArray] | None = None, always_exchange: bool = False):
        self.eng, self.cfg = engine, cfg
        self.comm = comm
        # always_exchange: run every collective even in a group of one (rehearses the RCCL calls on a
        # one-GPU box; a sum / min / broadcast over one rank leaves the buffers bit-identical)
        self.always_exchange = bool(always_exchange) and comm is not None
        self.world = comm.world if comm else 1
        self.n_total = cfg.n_total if cfg.n_total is not None else cfg.n_frames * self.world
        eng = engine
        F, d, k, n = cfg.n_features, cfg.cluster_dim, cfg.k, cfg.n_frames
        self.buf = shared if shared is not None else {
            nm: eng.zeros(shape, np.dtype(dt)) for nm, (shape, dt) in exchange_shapes(cfg).items()}
        b = self.buf
        for name, (parent, first, length) in exchange_aliases(cfg).items():
            if name not in b:
                b[name] = b[parent].view((length,), offset_elems=first)
        b["centers"] = b["start"].view((k, d))          # the kernels take the centres as a [k, d] table
        # featurize front stage: xyz stays resident, the feature matrix is rebuilt by every step
        if cfg.n_atoms > 0:
            if cfg.pairs is None or len(cfg.pairs) != F:
                raise ValueError("ShardConfig.pairs must list n_features atom pairs")
            self.xyz = x
            self.pairs_d = eng.to_device(np.ascontiguousarray(cfg.pairs, np.int32).reshape(F, 2))
            self.x = eng.empty((n, F), np.float32)
            eng.featurize_distances_into(self.xyz, self.pairs_d, self.x)
        else:
            self.xyz = None
            self.x = x
        if cfg.tica_dim > 0:
            self.mean, self.scale, self.inv_scale = (eng.empty((F,), np.float64) for _ in range(3))
            self.eig = eng.empty((F,), np.float64)
            self.W = eng.empty((F, F), np.float64)
            self.m2 = eng.empty((F,), np.float64)
            self.rank_d = eng.empty((1,), np.int32)
            self.Y = eng.empty((n, d), np.float64)
        else:
            self.Y = self.x
        self.labels = eng.empty((n,), np.int32)
        self.T = eng.empty((k, k), np.float64)
        self.rowsum = eng.empty((k,), np.float64)
        self.diag = eng.empty((1,), np.float64)
        self.km_sums = b["km_acc"].view((k * d,), np.int64)
        self.km_counts = b["km_acc"].view((k,), np.int64, offset_elems=k * d)
        # The member sums of the Lloyd passes are incremental: they persist over the iterations and only the frames
        # that changed centre move their contribution (64-bit fixed-point integers: the bits of a full
        # re-accumulation; after two passes < 10 % of the frames move).  With several shards the persistent copy is
        # local and the exchange buffer receives a fresh copy of it before every all-reduce.
        multi_shard = self.comm is not None and (self.comm.world > 1 or self.always_exchange)
        self.km_local = eng.empty((k * d + k,), np.int64) if multi_shard else None
        # bf16 frame images of the k-means filter: built once per step after the projection, read by all Lloyd
        # passes and the final assignment (None when d / k are outside the filter's range)
        nbytes = eng.kmeans_image_bytes(n, d)
        self.km_image = eng.empty((nbytes,), np.uint8) if nbytes else None
        self.accum_events: list = []
        self.stage_events: dict = {}
        self.time_accum = False
        self.time_stages = False
        if cfg.tica_dim > 0:
            # one shift vector shared by all shards (row 0 of rank 0's shard) so that the raw
            # moment sums add across ranks
            _, first = eng.column_moments_partial(self.x)
            b["shift"].copy_from_host(first.to_host())
            if comm and (comm.world > 1 or self.always_exchange):
                comm.broadcast("shift", 0)

    @property
    def collectives_per_step(self) -> int:
        return (3 if self.cfg.tica_dim > 0 else 2) + self.cfg.kmeans_iters

    def _stamp(self, name: str):
        if self.time_stages:
            ev = self.eng.event()
            ev.record()
            self.stage_events.setdefault(name, []).append(ev)

    def step(self) -> None:
        eng, cfg, b, comm = self.eng, self.cfg, self.buf, self.comm
        multi = comm is not None and (comm.world > 1 or self.always_exchange)
        F, d, k = cfg.n_features, cfg.cluster_dim, cfg.k
        self._stamp("begin")
        # 0. featurize: pair distances from the resident coordinates
        if self.xyz is not None:
            eng.featurize_distances_into(self.xyz, self.pairs_d, self.x)
            self._stamp("featurize")
        if cfg.tica_dim > 0:
            # 1. time-lagged raw moments about the shared shift (fp64 MFMA): the only pass over X before the
            #    projection -- the standardisation sums follow from them and 2 * lag edge frames
            eng.lagged_moments(self.x, cfg.lag, b["shift"], assume_finite=True, out=b["lagged"], symmetric=True)
            eng.moments_from_lagged(self.x, cfg.lag, b["shift"], b["lagged"], out=b["mom_sums"])
            if multi:
                comm.allreduce_sum("moments")
            eng.standardise_params(b["mom_sums"], b["shift"], F, float(self.n_total), True,
                                   out=(self.mean, self.scale, self.inv_scale))
            self._stamp("moments")
            # 2. TICA solve (every rank solves the same F x F problem on identical bits)
            eng.tica_solve(b["lagged"], F, scale=self.scale, epsilon=1e-6, kinetic_map=True,
                           out=(self.eig, self.W, self.m2, self.rank_d))
            self._stamp("tica_solve")
            # 3. projection; max |Y| (the fixed-point scale of the Lloyd sums needs it) falls out of the same pass
            # (x - shift) / sigma - m2: m2 is the symmetric mean about the SAME shift the moments used
            eng.project(self.x, b["shift"], self.inv_scale, self.W, d, mean2=self.m2, out=self.Y,
                        absmax=b["fit_state"].view((1,), offset_elems=2), assume_finite=True)
            self._stamp("project")
        # 4. k-means: fixed number of Lloyd iterations over all frames
        eng.kmeans_fit_begin(self.Y, k, seed=cfg.seed, n_total=self.n_total, tol2=0.0, centers=b["centers"],
                             state=b["fit_state"], absmax_ready=cfg.tica_dim > 0)
        if multi:
            # identical start on every rank in ONE collective: the MIN over [centres | scale] where every rank but 0
            # offers 0x7F7F... = 1.4e306 for the centres (finite coordinates pass through bit for bit) and its own
            # fixed-point scale (state = {scale, inv_scale, ...}; inv_scale follows as the reciprocal: 2^-e, exact)
            if comm.rank != 0:
                b["centers"].fill_bytes_(0x7F)
            b["start_scale"].copy_from(b["fit_scale"])
            comm.allreduce_min("start")
            b["fit_scale"].copy_from(b["start_scale"])
            comm.reciprocal("fit_inv_scale", "fit_scale")
        acc = self.km_local if self.km_local is not None else b["km_acc"]
        acc.zero_()
        acc_sums, acc_counts = acc.view((k * d,), np.int64), acc.view((k,), np.int64, offset_elems=k * d)
        self.labels.fill_bytes_(0xFF)          # the centre each frame is booked under: none yet
        if self.km_image is not None:
            eng.kmeans_pack(self.Y, image=self.km_image)
        for _ in range(cfg.kmeans_iters):
            if self.time_accum:
                e0, e1 = eng.event(), eng.event()
                e0.record()
            if not multi:
                # one shard: the accumulate launch closes the iteration itself (its last workgroup)
                eng.kmeans_lloyd_pass(self.Y, b["centers"], b["fit_state"], acc_sums, acc_counts,
                                      image=self.km_image, prev_labels=self.labels)
            else:
                eng.kmeans_accumulate(self.Y, b["centers"], b["fit_state"], acc_sums, acc_counts,
                                      image=self.km_image, prev_labels=self.labels)
            if self.time_accum:
                e1.record()
                self.accum_events.append((e0, e1))
            if multi:
                comm.allreduce_sum_from("km_acc", self.km_local)     # out of place: no copy in front of the collective
                eng.kmeans_update(self.km_sums, self.km_counts, b["centers"], b["fit_state"], clear=False)
        eng.kmeans_assign(self.Y, b["centers"], labels=self.labels, image=self.km_image)
        self._stamp("kmeans")
        # 5. lag-tau counts (a whole lag scan in one pass and ONE collective) + row-normalised transition matrix
        if cfg.lags:
            L = len(cfg.lags)
            eng.count_transitions_lagscan(self.labels, k, cfg.lags, out=b["counts"].view((L, k, k)),
                                          pairs=b["counts"].view((L,), offset_elems=L * k * k))
            if multi:
                comm.allreduce_sum("counts")
            # the matrix of the model lag (the first lag of the scan that equals cfg.lag, else the first one)
            li = list(cfg.lags).index(cfg.lag) if cfg.lag in cfg.lags else 0
            eng.row_normalise_into(b["counts"].view((k, k), offset_elems=li * k * k), self.T, self.rowsum, self.diag)
        else:
            eng.count_transitions(self.labels, k, cfg.lag, out=b["counts"].view((k, k)),
                                  pairs=b["counts"].view((1,), offset_elems=k * k))
            if multi:
                comm.allreduce_sum("counts")
            eng.row_normalise_into(b["counts"].view((k, k)), self.T, self.rowsum, self.diag)
        self._stamp("counts")

    def lagscan_counts(self) -> np.ndarray:
        """The (all-reduced) count matrices of the lag scan, [L, k, k] int64 on the host."""
        L, k = len(self.cfg.lags), self.cfg.k
        return self.buf["counts"].view((L, k, k)).to_host()


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

