"""Device context and buffers: the thin host-side layer over the C ABI.

``Engine`` owns one ``msm_ctx`` (one HIP device, one stream).  ``DeviceArray``
is a typed view of hipMalloc'ed memory; arrays can also wrap foreign device
pointers (e.g. ``torch.Tensor.data_ptr()``) so collectives can run on them.
"""

from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Iterable, Sequence

import numpy as np

from . import _lib
from ._lib import check, lib

__all__ = ["Engine", "DeviceArray", "get_engine", "segments_to_bounds"]


def _dtype_code(dtype: np.dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return _lib.MSM_F32
    if dtype == np.float64:
        return _lib.MSM_F64
    raise TypeError(f"feature matrices must be float32 or float64, got {dtype}")


def segments_to_bounds(
    segments: Iterable[tuple[int, int]] | None, n: int
) -> tuple[np.ndarray, np.ndarray]:
    """[(start, stop), ...] -> (starts, stops) int64, clipped like
    pmarlo.analysis.discretize._iter_segments (S/analysis/discretize.py:596-606)."""
    if segments is None:
        return np.zeros(1, np.int64), np.full(1, int(n), np.int64)
    starts, stops = [], []
    for start, stop in segments:
        a, b = max(0, int(start)), min(int(n), int(stop))
        if b > a:
            starts.append(a)
            stops.append(b)
    return np.asarray(starts, np.int64), np.asarray(stops, np.int64)


class DeviceArray:
    """A shaped, typed block of device memory."""

    __slots__ = ("engine", "ptr", "shape", "dtype", "_owned", "_base", "_alloc")

    def __init__(self, engine: "Engine", ptr: int, shape, dtype, owned: bool, alloc: int = 0):
        self.engine = engine
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self._owned = owned
        self._base = None
        self._alloc = int(alloc)      # bytes of the block behind an owned array (the engine's pool takes it back)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def nbytes(self) -> int:
        return self.size * self.dtype.itemsize

    def to_host(self) -> np.ndarray:
        out = np.empty(self.shape, self.dtype)
        if out.nbytes:
            check(lib.msm_memcpy_d2h(self.engine.handle, out.ctypes.data, self.ptr, out.nbytes),
                  self.engine.handle)
        return out

    def copy_from_host(self, arr: np.ndarray) -> "DeviceArray":
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        if arr.size != self.size:
            raise ValueError(f"size mismatch: device {self.shape} vs host {arr.shape}")
        if arr.nbytes:
            check(lib.msm_memcpy_h2d(self.engine.handle, self.ptr, arr.ctypes.data, arr.nbytes),
                  self.engine.handle)
        return self

    def copy_from(self, src: "DeviceArray") -> "DeviceArray":
        """Device-to-device copy of `src` (same byte count) into this array."""
        if src.nbytes != self.nbytes:
            raise ValueError(f"size mismatch: {self.nbytes} vs {src.nbytes} bytes")
        if self.nbytes:
            check(lib.msm_memcpy_d2d(self.engine.handle, self.ptr, src.ptr, self.nbytes), self.engine.handle)
        return self

    def zero_(self) -> "DeviceArray":
        check(lib.msm_memset(self.engine.handle, self.ptr, 0, self.nbytes), self.engine.handle)
        return self

    def fill_bytes_(self, byte: int) -> "DeviceArray":
        """Every byte set to `byte` (0xFF makes int32 / int64 entries -1)."""
        check(lib.msm_memset(self.engine.handle, self.ptr, int(byte) & 0xFF, self.nbytes), self.engine.handle)
        return self

    def view(self, shape, dtype=None, offset_elems: int = 0) -> "DeviceArray":
        dtype = self.dtype if dtype is None else np.dtype(dtype)
        v = DeviceArray(self.engine, self.ptr + offset_elems * self.dtype.itemsize, shape, dtype, False)
        v._base = self      # a view keeps the allocation it points into alive
        return v

    def free(self) -> None:
        if self._owned and self.ptr and self.engine.handle:
            self.engine._release(self.ptr, self._alloc)
        self.ptr = 0
        self._owned = False

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.free()
        except Exception:
            pass


class _Event:
    def __init__(self, engine: "Engine"):
        self.engine = engine
        h = C.c_void_p()
        check(lib.msm_event_create(engine.handle, C.byref(h)), engine.handle)
        self.handle = h

    def record(self) -> "_Event":
        check(lib.msm_event_record(self.engine.handle, self.handle), self.engine.handle)
        return self

    def elapsed_ms(self, stop: "_Event") -> float:
        ms = C.c_float()
        check(lib.msm_event_elapsed_ms(self.handle, stop.handle, C.byref(ms)), self.engine.handle)
        return float(ms.value)

    def __del__(self):  # pragma: no cover
        try:
            if self.handle:
                lib.msm_event_destroy(self.handle)
        except Exception:
            pass


class Engine:
    """One HIP device + stream, and the kernels of the MSM path on it."""

    def __init__(self, device: int = 0, stream: int | None = None):
        h = C.c_void_p()
        status = lib.msm_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if status != _lib.MSM_OK:
            raise _lib.MsmError(
                f"msm_ctx_create(device={device}) failed with status {status}: "
                "no usable HIP device (pmarlo_amd has no CPU fallback)"
            )
        self.handle = h
        self.device = int(device)
        # freed blocks wait here for the next request of their size: hipMalloc / hipFree synchronise the device and
        # cost 0.1-1 ms each, which was a third of an operator call (22 frees = 10 ms in discretize_dataset at 1 M
        # frames).  All work of an engine is ordered on its one stream, so a block can be handed out again at once.
        self._pool: dict[int, list[int]] = {}
        self._pool_bytes = 0
        self._pool_cap = int(os.environ.get("MSM_POOL_BYTES", str(16 << 30)))

    # -- plumbing -----------------------------------------------------------
    @staticmethod
    def _block_size(nbytes: int) -> int:
        nbytes = max(int(nbytes), 1)
        step = 512 if nbytes < (1 << 20) else (1 << 20)
        return -(-nbytes // step) * step

    def _release(self, ptr: int, alloc: int) -> None:
        if alloc and self._pool_bytes + alloc <= self._pool_cap:
            self._pool.setdefault(alloc, []).append(ptr)
            self._pool_bytes += alloc
        else:
            lib.msm_free(self.handle, ptr)

    def empty_cache(self) -> None:
        """Give the pooled blocks back to the driver."""
        for blocks in self._pool.values():
            for ptr in blocks:
                lib.msm_free(self.handle, ptr)
        self._pool.clear()
        self._pool_bytes = 0

    def close(self) -> None:
        if self.handle:
            self.empty_cache()
            lib.msm_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        arch = C.create_string_buffer(64)
        ncu = C.c_int()
        mem = C.c_size_t()
        check(lib.msm_device_info(self.handle, arch, 64, C.byref(ncu), C.byref(mem)), self.handle)
        return {"arch": arch.value.decode(), "n_cu": int(ncu.value), "total_mem": int(mem.value)}

    def empty(self, shape, dtype) -> DeviceArray:
        dtype = np.dtype(dtype)
        shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)
        nbytes = int(np.prod(shape)) * dtype.itemsize if shape else dtype.itemsize
        size = self._block_size(nbytes)
        blocks = self._pool.get(size)
        if blocks:
            self._pool_bytes -= size
            return DeviceArray(self, blocks.pop(), shape, dtype, True, size)
        p = C.c_void_p()
        status = lib.msm_malloc(self.handle, size, C.byref(p))
        if status != _lib.MSM_OK and self._pool_bytes:
            self.empty_cache()          # out of memory with blocks parked in the pool: hand them back and try again
            status = lib.msm_malloc(self.handle, size, C.byref(p))
        check(status, self.handle)
        return DeviceArray(self, p.value, shape, dtype, True, size)

    def zeros(self, shape, dtype) -> DeviceArray:
        return self.empty(shape, dtype).zero_()

    def to_device(self, arr: np.ndarray, dtype=None) -> DeviceArray:
        arr = np.ascontiguousarray(arr, dtype=dtype)
        return self.empty(arr.shape, arr.dtype).copy_from_host(arr)

    def wrap(self, ptr: int, shape, dtype) -> DeviceArray:
        """View foreign device memory (torch tensor data_ptr, etc.)."""
        return DeviceArray(self, ptr, shape, dtype, False)

    def sync(self) -> None:
        check(lib.msm_sync(self.handle), self.handle)

    def event(self) -> _Event:
        return _Event(self)

    def graph_begin(self) -> None:
        check(lib.msm_graph_begin(self.handle), self.handle)

    def graph_end(self):
        g = C.c_void_p()
        check(lib.msm_graph_end(self.handle, C.byref(g)), self.handle)
        return g

    def graph_launch(self, g) -> None:
        check(lib.msm_graph_launch(self.handle, g), self.handle)

    def graph_destroy(self, g) -> None:
        lib.msm_graph_destroy(g)

    # -- featurizers -----------------------------------------------------------
    def featurize_rg(self, xyz: DeviceArray) -> DeviceArray:
        """Radius of gyration (unit masses) per frame: float32 [n]."""
        n, A, _ = xyz.shape
        out = self.empty((n,), np.float32)
        check(lib.msm_featurize_rg(self.handle, xyz.ptr, n, A, out.ptr, 1, 0), self.handle)
        return out

    def featurize_contacts(self, xyz: DeviceArray, pairs, rcut: float) -> DeviceArray:
        """1.0 where the pair distance is <= rcut (nm): float32 [n, P]."""
        n, A, _ = xyz.shape
        p = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        if p.size and (p.min() < 0 or p.max() >= A):
            raise ValueError(f"atom index out of range [0, {A})")
        pd_ = self.to_device(p)
        out = self.empty((n, p.shape[0]), np.float32)
        check(lib.msm_featurize_contacts(self.handle, xyz.ptr, n, A, pd_.ptr, p.shape[0], float(rcut), out.ptr,
                                         p.shape[0], 0), self.handle)
        self.sync()
        return out

    def featurize_sasa(self, xyz: DeviceArray, radii, points) -> DeviceArray:
        """Shrake-Rupley accessible area per atom: float32 [n, A] (nm^2).  `radii` float32 [A] already include the
        probe radius, `points` float32 [P, 3] are the unit sphere points."""
        n, A, _ = xyz.shape
        r = self.to_device(np.ascontiguousarray(radii, np.float32).reshape(A))
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        p = self.to_device(pts)
        out = self.empty((n, A), np.float32)
        check(lib.msm_featurize_sasa(self.handle, xyz.ptr, n, A, r.ptr, p.ptr, pts.shape[0], out.ptr), self.handle)
        return out

    def hbond_presence(self, xyz: DeviceArray, triplets, dist_cutoff: float, angle_cutoff: float) -> np.ndarray:
        """Frames in which each (donor, hydrogen, acceptor) triplet meets the Baker-Hubbard criteria: int64 [Tn]."""
        n, A, _ = xyz.shape
        tr = np.ascontiguousarray(triplets, np.int32).reshape(-1, 3)
        if tr.size and (tr.min() < 0 or tr.max() >= A):
            raise ValueError(f"atom index out of range [0, {A})")
        if tr.shape[0] == 0:
            return np.zeros((0,), np.int64)
        td = self.to_device(tr)
        counts = self.empty((tr.shape[0],), np.uint64)
        check(lib.msm_hbond_presence(self.handle, xyz.ptr, n, A, td.ptr, tr.shape[0], float(dist_cutoff),
                                     float(angle_cutoff), counts.ptr), self.handle)
        return counts.to_host().astype(np.int64)

    def dssp(self, xyz: DeviceArray, backbone, chain, proline) -> np.ndarray:
        """Kabsch-Sander codes per frame and protein residue: uint8 [n, R] (0 loop, 1 H, 2 B, 3 E, 4 G, 5 I, 6 T, 7 S)."""
        n, A, _ = xyz.shape
        bb = np.ascontiguousarray(backbone, np.int32).reshape(-1, 4)
        R = bb.shape[0]
        if R == 0 or n == 0:
            return np.zeros((n, R), np.uint8)
        if bb.min() < 0 or bb.max() >= A:
            raise ValueError(f"atom index out of range [0, {A})")
        bd = self.to_device(bb)
        cd = self.to_device(np.ascontiguousarray(chain, np.int32).reshape(R))
        pd_ = self.to_device(np.ascontiguousarray(proline, np.uint8).reshape(R))
        codes = self.empty((n, R), np.uint8)
        check(lib.msm_dssp(self.handle, xyz.ptr, n, A, bd.ptr, cd.ptr, pd_.ptr, R, codes.ptr), self.handle)
        return codes.to_host()

    def featurize(self, xyz: DeviceArray, *, pairs=None, triplets=None, quads=None, dihedral_mode: int = 0,
                  out: DeviceArray | None = None) -> DeviceArray:
        """xyz float32 [n, A, 3] -> float32 [n, F]: distance columns, then angle columns, then
        dihedral columns (mode 0 radians, 1 interleaved cos/sin, 2 cos block | sin block)."""
        if xyz.dtype != np.float32 or len(xyz.shape) != 3 or xyz.shape[2] != 3:
            raise ValueError("xyz must be float32 with shape (n_frames, n_atoms, 3)")
        n, A, _ = xyz.shape

        def _idx(arr, width):
            if arr is None:
                return None, 0
            a = np.ascontiguousarray(arr, np.int32).reshape(-1, width)
            if a.size and (a.min() < 0 or a.max() >= A):
                raise ValueError(f"atom index out of range [0, {A})")
            return (self.to_device(a) if a.size else None), a.shape[0]

        dp, P = _idx(pairs, 2)
        dt, Tn = _idx(triplets, 3)
        dq, Q = _idx(quads, 4)
        width = P + Tn + (Q if dihedral_mode == 0 else 2 * Q)
        out = out if out is not None else self.empty((n, width), np.float32)
        ld = out.shape[1]
        if P:
            check(lib.msm_featurize_distances(self.handle, xyz.ptr, n, A, dp.ptr, P, out.ptr, ld, 0), self.handle)
        if Tn:
            check(lib.msm_featurize_angles(self.handle, xyz.ptr, n, A, dt.ptr, Tn, out.ptr, ld, P), self.handle)
        if Q:
            check(lib.msm_featurize_dihedrals(self.handle, xyz.ptr, n, A, dq.ptr, Q, int(dihedral_mode), out.ptr, ld,
                                              P + Tn), self.handle)
        self.sync()  # index tables are freed on return
        return out

    def featurize_distances_into(self, xyz: DeviceArray, pairs: DeviceArray, out: DeviceArray, col0: int = 0) -> DeviceArray:
        """Pair distances with a device-resident int32 [P, 2] index table into columns col0.. of `out` (float32
        [n, ld]); no upload, no sync: the form a timed step uses."""
        n, A, _ = xyz.shape
        check(lib.msm_featurize_distances(self.handle, xyz.ptr, n, A, pairs.ptr, pairs.shape[0], out.ptr, out.shape[1],
                                          int(col0)), self.handle)
        return out

    # -- transition counts ----------------------------------------------------
    @staticmethod
    def _seg_ptrs(starts: np.ndarray, stops: np.ndarray):
        starts = np.ascontiguousarray(starts, np.int64)
        stops = np.ascontiguousarray(stops, np.int64)
        if starts.shape != stops.shape:
            raise ValueError("segment starts/stops differ in length")
        return starts, stops

    def count_transitions(self, labels: DeviceArray, k: int, lag: int, *, starts=None, stops=None,
                          stride: int = 1, out: DeviceArray | None = None,
                          pairs: DeviceArray | None = None) -> tuple[DeviceArray, DeviceArray]:
        n = labels.size
        if starts is None:
            starts, stops = segments_to_bounds(None, n)
        starts, stops = self._seg_ptrs(starts, stops)
        out = out if out is not None else self.empty((k, k), np.int64)
        pairs = pairs if pairs is not None else self.empty((1,), np.int64)
        check(lib.msm_count_transitions(self.handle, labels.ptr, n, starts.ctypes.data, stops.ctypes.data,
                                        len(starts), int(lag), int(stride), int(k), out.ptr, pairs.ptr),
              self.handle)
        return out, pairs

    def count_transitions_weighted(self, labels: DeviceArray, weights: DeviceArray, k: int, lag: int, *,
                                   starts=None, stops=None, stride: int = 1,
                                   out: DeviceArray | None = None, pairs: DeviceArray | None = None):
        n = labels.size
        if weights.size != n or weights.dtype != np.float64:
            raise ValueError("weights must be float64 with one entry per label")
        if starts is None:
            starts, stops = segments_to_bounds(None, n)
        starts, stops = self._seg_ptrs(starts, stops)
        out = out if out is not None else self.empty((k, k), np.float64)
        pairs = pairs if pairs is not None else self.empty((1,), np.int64)
        check(lib.msm_count_transitions_weighted(self.handle, labels.ptr, weights.ptr, n, starts.ctypes.data,
                                                 stops.ctypes.data, len(starts), int(lag), int(stride),
                                                 int(k), out.ptr, pairs.ptr), self.handle)
        return out, pairs

    def count_transitions_lagscan(self, labels: DeviceArray, k: int, lags: Sequence[int], *, starts=None,
                                  stops=None, out: DeviceArray | None = None,
                                  pairs: DeviceArray | None = None):
        n = labels.size
        lags_arr = np.ascontiguousarray(lags, np.int32)
        if starts is None:
            starts, stops = segments_to_bounds(None, n)
        starts, stops = self._seg_ptrs(starts, stops)
        L = len(lags_arr)
        out = out if out is not None else self.empty((L, k, k), np.int64)
        pairs = pairs if pairs is not None else self.empty((L,), np.int64)
        check(lib.msm_count_transitions_lagscan(self.handle, labels.ptr, n, starts.ctypes.data,
                                                stops.ctypes.data, len(starts), lags_arr.ctypes.data, L,
                                                int(k), out.ptr, pairs.ptr), self.handle)
        return out, pairs

    def state_counts(self, labels: DeviceArray, k: int, out: DeviceArray | None = None) -> DeviceArray:
        out = out if out is not None else self.empty((k,), np.int64)
        check(lib.msm_state_counts(self.handle, labels.ptr, labels.size, int(k), out.ptr), self.handle)
        return out

    def run_lengths(self, labels: DeviceArray, k: int):
        """Dwell-time runs of a label sequence (negative labels separate trajectories) ->
        (stats int64 [4, k] = min / max / sum / number per state, run states int32 [R], run lengths int64 [R])."""
        n = labels.size
        stats = self.empty((4, int(k)), np.int64)
        rs, rl = self.empty((max(n, 1),), np.int32), self.empty((max(n, 1),), np.int64)
        cnt = self.empty((1,), np.int64)
        check(lib.msm_run_lengths(self.handle, labels.ptr, n, int(k), stats.ptr, rs.ptr, rl.ptr, n, cnt.ptr), self.handle)
        r = int(cnt.to_host()[0])
        return stats.to_host(), rs.to_host()[:r], rl.to_host()[:r]

    # -- moments / covariance / TICA -------------------------------------------
    def column_moments(self, x: DeviceArray, ddof: int = 0):
        """-> (mean, std, count) device arrays [F] f64 (NaN entries skipped)."""
        n, F = x.shape
        mean, std, cnt = (self.empty((F,), np.float64) for _ in range(3))
        check(lib.msm_column_moments(self.handle, x.ptr, _dtype_code(x.dtype), n, F, F, int(ddof), mean.ptr,
                                     std.ptr, cnt.ptr), self.handle)
        return mean, std, cnt

    def column_minmax(self, x: DeviceArray):
        """(min [F], max [F] over the finite entries, int64 [2] = non-finite entries, fully finite rows)."""
        n, F = x.shape
        mn, mx, cnt = self.empty((F,), np.float64), self.empty((F,), np.float64), self.empty((2,), np.int64)
        check(lib.msm_column_minmax(self.handle, x.ptr, _dtype_code(x.dtype), n, F, F, mn.ptr, mx.ptr, cnt.ptr), self.handle)
        return mn, mx, cnt

    def column_moments_partial(self, x: DeviceArray, shift: DeviceArray | None = None,
                               sums: DeviceArray | None = None):
        """Raw sums [cnt | S1 | S2] (3F f64) about `shift` (default: row 0) -> (sums, shift)."""
        n, F = x.shape
        sums = sums if sums is not None else self.empty((3 * F,), np.float64)
        shift_out = self.empty((F,), np.float64) if shift is None else None  # no allocation in steady state
        check(lib.msm_column_moments_partial(self.handle, x.ptr, _dtype_code(x.dtype), n, F, F,
                                             shift.ptr if shift is not None else None, sums.ptr,
                                             shift_out.ptr if shift_out is not None else None), self.handle)
        return sums, (shift_out if shift_out is not None else shift)

    def moments_finalize(self, sums: DeviceArray, shift: DeviceArray, F: int, ddof: int = 0):
        mean, std, cnt = (self.empty((F,), np.float64) for _ in range(3))
        check(lib.msm_moments_finalize(self.handle, sums.ptr, shift.ptr, F, int(ddof), mean.ptr, std.ptr,
                                       cnt.ptr), self.handle)
        return mean, std, cnt

    def standardise_params(self, sums: DeviceArray, shift: DeviceArray, F: int, n_rows: float, with_std: bool = True,
                           out=None):
        """-> (mean, scale, inv_scale) device arrays [F]: the _preprocess parameters, no host sync."""
        mean, scale, inv = out if out is not None else tuple(self.empty((F,), np.float64) for _ in range(3))
        check(lib.msm_standardise_params(self.handle, sums.ptr, shift.ptr, F, float(n_rows), int(bool(with_std)),
                                         mean.ptr, scale.ptr, inv.ptr), self.handle)
        return mean, scale, inv

    def lagged_moments(self, x: DeviceArray, lag: int, shift: DeviceArray, *, starts=None, stops=None,
                       assume_finite: bool = False, out: DeviceArray | None = None,
                       one_sided: bool = False, symmetric: bool = False) -> DeviceArray:
        """Raw lagged moments [M00 | M0t | sx | sy | T] (2F^2+2F+1 f64): the reversible estimator's
        (M00 over X0 and Yt) or, with one_sided, M00 over X0 only.  symmetric: the M0t block holds
        (M0t + M0t') / 2 -- all a reversible TICA reads of it -- from the cheaper symmetric accumulation."""
        if one_sided and symmetric:
            raise ValueError("one_sided and symmetric moments exclude each other")
        n, F = x.shape
        if starts is None:
            starts, stops = segments_to_bounds(None, n)
        starts, stops = self._seg_ptrs(starts, stops)
        out = out if out is not None else self.empty((2 * F * F + 2 * F + 1,), np.float64)
        fn = (lib.msm_lagged_moments_onesided if one_sided
              else lib.msm_lagged_moments_reversible if symmetric else lib.msm_lagged_moments)
        check(fn(self.handle, x.ptr, _dtype_code(x.dtype), n, F, F, starts.ctypes.data, stops.ctypes.data, len(starts),
                 int(lag), shift.ptr, int(bool(assume_finite)), out.ptr), self.handle)
        return out

    def onesided_tica_eigenvalues(self, moments: DeviceArray, F: int, clip: float = 1e-12) -> DeviceArray:
        """Descending eigenvalues of the reference's in-repo estimator from one-sided moments."""
        out = self.empty((F,), np.float64)
        check(lib.msm_onesided_tica_eigenvalues(self.handle, moments.ptr, int(F), float(clip), out.ptr), self.handle)
        return out

    def moments_from_lagged(self, x: DeviceArray, lag: int, shift: DeviceArray, moments: DeviceArray, *, starts=None,
                            stops=None, out: DeviceArray | None = None) -> DeviceArray:
        """[cnt | S1 | S2] over all frames from lagged moments of the same shift (finite data only)."""
        n, F = x.shape
        if starts is None:
            starts, stops = segments_to_bounds(None, n)
        starts, stops = self._seg_ptrs(starts, stops)
        out = out if out is not None else self.empty((3 * F,), np.float64)
        check(lib.msm_moments_from_lagged(self.handle, x.ptr, _dtype_code(x.dtype), n, F, F, starts.ctypes.data,
                                          stops.ctypes.data, len(starts), int(lag), shift.ptr, moments.ptr, out.ptr),
              self.handle)
        return out

    def tica_solve(self, moments: DeviceArray, F: int, *, scale: DeviceArray | None = None,
                   epsilon: float = 1e-6, kinetic_map: bool = True, out=None):
        """-> (eigvals [F], coeffs [F,F], mean [F], rank int32[1]) on the device (`out`: the same four, preallocated)."""
        if out is not None:
            eig, W, mean, rank = out
        else:
            eig = self.empty((F,), np.float64)
            W = self.empty((F, F), np.float64)
            mean = self.empty((F,), np.float64)
            rank = self.empty((1,), np.int32)
        check(lib.msm_tica_solve(self.handle, moments.ptr, scale.ptr if scale is not None else None, F,
                                 float(epsilon), int(bool(kinetic_map)), eig.ptr, W.ptr, mean.ptr, rank.ptr),
              self.handle)
        return eig, W, mean, rank

    def project(self, x: DeviceArray, mu: DeviceArray, inv_sigma: DeviceArray, W: DeviceArray, d: int, *,
                mean2: DeviceArray | None = None, out: DeviceArray | None = None,
                absmax: DeviceArray | None = None, assume_finite: bool = False) -> DeviceArray:
        """`absmax` (one f64, e.g. slot 2 of a k-means fit state) receives max |Y| from the same pass.
        assume_finite: X holds no NaN (column_moments' count tells); the per-element NaN test is left out."""
        n, F = x.shape
        ldw = W.shape[1]
        out = out if out is not None else self.empty((n, d), np.float64)
        fn = lib.msm_project_finite if assume_finite else lib.msm_project
        check(fn(self.handle, x.ptr, _dtype_code(x.dtype), n, F, F, mu.ptr, inv_sigma.ptr,
                 mean2.ptr if mean2 is not None else None, W.ptr, int(d), ldw, out.ptr,
                 out.shape[1], absmax.ptr if absmax is not None else None), self.handle)
        return out

    def eigh(self, a: DeviceArray, want_vectors: bool = True):
        n = a.shape[0]
        w = self.empty((n,), np.float64)
        v = self.empty((n, n), np.float64) if want_vectors else None
        sweeps = self.empty((1,), np.int32)
        check(lib.msm_eigh(self.handle, a.ptr, n, w.ptr, v.ptr if v is not None else None, sweeps.ptr),
              self.handle)
        return w, v, sweeps

    # -- k-means --------------------------------------------------------------
    def kmeans_assign(self, x: DeviceArray, centers: DeviceArray, *, mean: DeviceArray | None = None,
                      std: DeviceArray | None = None, labels: DeviceArray | None = None,
                      mindist: DeviceArray | None = None, image: DeviceArray | None = None) -> DeviceArray:
        n, d = x.shape
        k, dc = centers.shape
        if dc != d:
            raise ValueError(f"centres have {dc} features, data has {d}")
        if centers.dtype != np.float64:
            raise TypeError("centres must be float64")
        labels = labels if labels is not None else self.empty((n,), np.int32)
        check(lib.msm_kmeans_assign_packed(self.handle, x.ptr, _dtype_code(x.dtype), n, d, d, centers.ptr, k,
                                           mean.ptr if mean is not None else None,
                                           std.ptr if std is not None else None,
                                           image.ptr if image is not None else None, labels.ptr,
                                           mindist.ptr if mindist is not None else None), self.handle)
        return labels


    FIT_STATE_FIELDS = ("scale", "inv_scale", "absmax", "shift2", "tol2", "done", "n_iter", "inertia")

    def kmeans_fit(self, x: DeviceArray, k: int, *, seed: int = 0, max_iter: int = 50, tol2: float = 0.0,
                   mean: DeviceArray | None = None, std: DeviceArray | None = None,
                   centers: DeviceArray | None = None) -> tuple[DeviceArray, DeviceArray]:
        """Lloyd k-means on one device -> (centers [k,d] f64, state [8] f64).
        Pass `centers` to start from given centres instead of the seeded stratified draw."""
        n, d = x.shape
        init = centers is None
        centers = centers if centers is not None else self.empty((k, d), np.float64)
        state = self.zeros((8,), np.float64)
        check(lib.msm_kmeans_fit(self.handle, x.ptr, _dtype_code(x.dtype), n, d, d,
                                 mean.ptr if mean is not None else None, std.ptr if std is not None else None,
                                 int(k), int(seed) & (2**64 - 1), int(init), int(max_iter), float(tol2), centers.ptr,
                                 state.ptr), self.handle)
        return centers, state

    def kmeans_init_plusplus(self, x: DeviceArray, k: int, *, seed: int = 0, mean: DeviceArray | None = None,
                             std: DeviceArray | None = None, n_total: int | None = None,
                             centers: DeviceArray | None = None, return_picked: bool = False):
        """k-means++ seeding on the device (csrc/kmeanspp.hip): centres [k, d] f64 = frames drawn with probability
        proportional to the squared distance to the nearest centre already drawn (deterministic in the data and seed)."""
        n, d = x.shape
        n_total = n if n_total is None else int(n_total)
        centers, state = self.kmeans_fit_begin(x, k, seed=seed, n_total=n_total, tol2=0.0, mean=mean, std=std,
                                               centers=centers, init_centers=False)
        picked = self.empty((k,), np.int64) if return_picked else None
        check(lib.msm_kmeans_init_plusplus(self.handle, x.ptr, _dtype_code(x.dtype), n, d, d,
                                           mean.ptr if mean is not None else None, std.ptr if std is not None else None,
                                           int(k), int(seed) & (2**64 - 1), float(n_total), centers.ptr, state.ptr,
                                           picked.ptr if picked is not None else None), self.handle)
        return (centers, picked) if return_picked else centers

    def kmeans_fit_minibatch(self, x: DeviceArray, k: int, *, seed: int = 0, batch_size: int = 100, max_iter: int = 5,
                             init: str = "kmeans++", centers: DeviceArray | None = None,
                             max_batches: int = 4096) -> tuple[DeviceArray, DeviceArray]:
        """Mini-batch k-means (Sculley; what sklearn's MiniBatchKMeans and deeptime's do with their batches): the
        frames are dealt into `nb` interleaved batches (batch b = frames b, b + nb, ...: a batch of a time series should
        not be one stretch of it), every batch is assigned to the current centres and ADDED to the running member sums,
        and the centres are the running means -- `max_iter` sweeps over all batches.  nb = ceil(n / batch_size), at
        most `max_batches` (two launches per batch).  Same kernels as the full-batch fit (msm_kmeans_accumulate /
        msm_kmeans_update through their row stride), so the sums are exact fixed point and the result is
        deterministic."""
        n, d = x.shape
        if batch_size < 1 or max_iter < 0:
            raise ValueError("batch_size must be >= 1 and max_iter >= 0")
        given = centers is not None
        # the running sums see every frame once per sweep: the fixed-point scale is sized for n * max_iter terms
        n_terms = n * max(1, int(max_iter))
        centers, state = self.kmeans_fit_begin(x, k, seed=seed, n_total=n_terms, tol2=0.0, centers=centers,
                                               init_centers=not given and init != "kmeans++")
        if not given and init == "kmeans++":
            check(lib.msm_kmeans_init_plusplus(self.handle, x.ptr, _dtype_code(x.dtype), n, d, d, None, None, int(k),
                                               int(seed) & (2**64 - 1), float(n_terms), centers.ptr, state.ptr, None),
                  self.handle)
        nb = max(1, min(int(max_batches), -(-n // int(batch_size))))
        sums, counts = self.zeros((k * d,), np.int64), self.zeros((k,), np.int64)
        item = x.dtype.itemsize
        for _ in range(int(max_iter)):
            for b in range(nb):
                rows = (n - b + nb - 1) // nb
                if rows <= 0:
                    continue
                check(lib.msm_kmeans_accumulate(self.handle, x.ptr + b * d * item, _dtype_code(x.dtype), rows, d, nb * d,
                                                centers.ptr, k, None, None, state.ptr, sums.ptr, counts.ptr), self.handle)
                check(lib.msm_kmeans_update(self.handle, sums.ptr, counts.ptr, k, d, centers.ptr, state.ptr, 0), self.handle)
        return centers, state

    def kmeans_fit_begin(self, x: DeviceArray, k: int, *, seed: int, n_total: int, tol2: float,
                         mean: DeviceArray | None = None, std: DeviceArray | None = None,
                         centers: DeviceArray | None = None, init_centers: bool = True,
                         state: DeviceArray | None = None, absmax_ready: bool = False):
        """absmax_ready: slot 2 of `state` already holds max |x| (left there by project(absmax=...))."""
        n, d = x.shape
        centers = centers if centers is not None else self.empty((k, d), np.float64)
        state = state if state is not None else self.zeros((8,), np.float64)
        check(lib.msm_kmeans_fit_begin(self.handle, x.ptr, _dtype_code(x.dtype), n, d, d,
                                       mean.ptr if mean is not None else None,
                                       std.ptr if std is not None else None, int(k), int(seed) & (2**64 - 1),
                                       int(bool(init_centers)), float(n_total), float(tol2), centers.ptr, state.ptr,
                                       int(bool(absmax_ready))), self.handle)
        return centers, state

    def kmeans_accumulate(self, x: DeviceArray, centers: DeviceArray, state: DeviceArray, sums: DeviceArray,
                          counts: DeviceArray, *, mean: DeviceArray | None = None, std: DeviceArray | None = None,
                          image: DeviceArray | None = None, prev_labels: DeviceArray | None = None):
        """Member sums / counts of one Lloyd pass.  With `prev_labels` (int32 [n], -1 before the first pass, updated
        in place) only the frames that changed centre move their contribution: `sums` / `counts` must then persist
        between the passes (kmeans_update(clear=False))."""
        n, d = x.shape
        k = centers.shape[0]
        if prev_labels is not None:
            check(lib.msm_kmeans_accumulate_delta(self.handle, x.ptr, _dtype_code(x.dtype), n, d, d, centers.ptr, k,
                                                  mean.ptr if mean is not None else None,
                                                  std.ptr if std is not None else None,
                                                  image.ptr if image is not None else None, state.ptr,
                                                  prev_labels.ptr, sums.ptr, counts.ptr), self.handle)
            return
        check(lib.msm_kmeans_accumulate_packed(self.handle, x.ptr, _dtype_code(x.dtype), n, d, d, centers.ptr, k,
                                               mean.ptr if mean is not None else None,
                                               std.ptr if std is not None else None,
                                               image.ptr if image is not None else None, state.ptr, sums.ptr,
                                               counts.ptr), self.handle)

    # -- bf16 frame images of the certified filter (msm_kmeans_pack) -----------------------------------------
    def kmeans_image_bytes(self, n: int, d: int) -> int:
        out = C.c_size_t(0)
        check(lib.msm_kmeans_image_bytes(int(n), int(d), C.byref(out)), self.handle)
        return int(out.value)

    def kmeans_pack(self, x: DeviceArray, *, mean: DeviceArray | None = None, std: DeviceArray | None = None,
                    image: DeviceArray | None = None) -> DeviceArray | None:
        """Frame images for repeated k-means passes over the same frames; None when d is outside the filter's range."""
        n, d = x.shape
        nbytes = self.kmeans_image_bytes(n, d)
        if nbytes == 0:
            return None
        image = image if image is not None else self.empty((nbytes,), np.uint8)
        if image.nbytes < nbytes:
            raise ValueError(f"image buffer holds {image.nbytes} bytes, {nbytes} needed")
        check(lib.msm_kmeans_pack(self.handle, x.ptr, _dtype_code(x.dtype), n, d, d,
                                  mean.ptr if mean is not None else None, std.ptr if std is not None else None,
                                  image.ptr), self.handle)
        return image

    def kmeans_filter_scanned(self, reset: bool = False) -> int:
        """Frames that took the filter's exhaustive fp64 scan since the context was created (diagnostics)."""
        out = C.c_uint64(0)
        check(lib.msm_kmeans_filter_scanned(self.handle, C.byref(out), int(reset)), self.handle)
        return int(out.value)

    def mfma_bf16_probe(self, a_bits: np.ndarray, b_bits: np.ndarray, c: np.ndarray) -> np.ndarray:
        """D = A B + C through one v_mfma_f32_16x16x32_bf16 per tile: a_bits [t, 16, 32], b_bits [t, 32, 16] uint16
        (raw bf16 patterns), c [t, 16, 16] float32 (host arrays).  The hardware rule behind the k-means filter."""
        a = np.ascontiguousarray(a_bits, np.uint16)
        b = np.ascontiguousarray(b_bits, np.uint16)
        cc = np.ascontiguousarray(c, np.float32)
        t = a.shape[0]
        if a.shape != (t, 16, 32) or b.shape != (t, 32, 16) or cc.shape != (t, 16, 16):
            raise ValueError("mfma_bf16_probe: shapes must be [t,16,32], [t,32,16], [t,16,16]")
        out = np.empty((t, 16, 16), np.float32)
        check(lib.msm_mfma_bf16_probe(self.handle, a.ctypes.data, b.ctypes.data, cc.ctypes.data, out.ctypes.data, t),
              self.handle)
        return out

    def kmeans_lloyd_pass(self, x: DeviceArray, centers: DeviceArray, state: DeviceArray, sums: DeviceArray,
                          counts: DeviceArray, *, mean: DeviceArray | None = None, std: DeviceArray | None = None,
                          image: DeviceArray | None = None, prev_labels: DeviceArray | None = None):
        """One Lloyd iteration on a single shard: kmeans_accumulate + kmeans_update(clear=False), in one launch where the
        filter kernel runs (msm_kmeans_lloyd_pass)."""
        n, d = x.shape
        k = centers.shape[0]
        check(lib.msm_kmeans_lloyd_pass(self.handle, x.ptr, _dtype_code(x.dtype), n, d, d, centers.ptr, k,
                                        mean.ptr if mean is not None else None, std.ptr if std is not None else None,
                                        image.ptr if image is not None else None, state.ptr,
                                        prev_labels.ptr if prev_labels is not None else None, sums.ptr, counts.ptr),
              self.handle)

    def kmeans_update(self, sums: DeviceArray, counts: DeviceArray, centers: DeviceArray, state: DeviceArray,
                      clear: bool = True):
        k, d = centers.shape
        check(lib.msm_kmeans_update(self.handle, sums.ptr, counts.ptr, k, d, centers.ptr, state.ptr, int(clear)),
              self.handle)

    def sum_f64(self, v: DeviceArray, out: DeviceArray | None = None) -> DeviceArray:
        out = out if out is not None else self.empty((1,), np.float64)
        check(lib.msm_sum_f64(self.handle, v.ptr, v.size, out.ptr), self.handle)
        return out


    # -- MSM estimation ---------------------------------------------------------
    def transition_matrix(self, counts: DeviceArray, *, mode: int = 0, alpha: float = 1e-3,
                          epsilon: float = 1e-12) -> dict:
        """counts [k,k] (int64 or float64) -> dict of device arrays (see msm_transition_matrix)."""
        k = counts.shape[0]
        if counts.dtype not in (np.dtype(np.int64), np.dtype(np.float64)):
            raise TypeError("counts must be int64 or float64")
        T = self.empty((k, k), np.float64)
        rowsum = self.empty((k,), np.float64)
        out = {"T": T, "rowsum": rowsum}
        if mode == 0:
            dm = self.empty((1,), np.float64)
            check(lib.msm_transition_matrix(self.handle, counts.ptr, int(counts.dtype == np.float64), k, 0, 0.0, 0.0,
                                            T.ptr, None, None, None, rowsum.ptr, dm.ptr), self.handle)
            out["diag_mass"] = dm
        else:
            active, inv = self.empty((k,), np.int32), self.empty((k,), np.int32)
            na = self.empty((1,), np.int32)
            check(lib.msm_transition_matrix(self.handle, counts.ptr, int(counts.dtype == np.float64), k, 1,
                                            float(alpha), float(epsilon), T.ptr, active.ptr, inv.ptr, na.ptr,
                                            rowsum.ptr, None), self.handle)
            out.update(active=active, inv_map=inv, n_active=na)
        return out

    def row_normalise_into(self, counts: DeviceArray, T: DeviceArray, rowsum: DeviceArray, diag_mass: DeviceArray):
        """T = counts / rowsum (zero rows stay zero), trace(T)/k: msm_transition_matrix mode 0 into given arrays."""
        k = counts.shape[0]
        check(lib.msm_transition_matrix(self.handle, counts.ptr, int(counts.dtype == np.float64), k, 0, 0.0, 0.0,
                                        T.ptr, None, None, None, rowsum.ptr, diag_mass.ptr), self.handle)

    def rcp(self, src: DeviceArray, dst: DeviceArray) -> None:
        check(lib.msm_rcp_f64(self.handle, src.ptr, src.size, dst.ptr), self.handle)

    def embed_full(self, T_active: DeviceArray, inv_map: DeviceArray, pi_active: DeviceArray | None = None):
        k = T_active.shape[0]
        T_full = self.empty((k, k), np.float64)
        pi_full = self.empty((k,), np.float64)
        check(lib.msm_embed_full(self.handle, T_active.ptr, pi_active.ptr if pi_active is not None else None,
                                 inv_map.ptr, k, T_full.ptr, pi_full.ptr), self.handle)
        return T_full, pi_full

    def silhouette(self, x_sorted: DeviceArray, offsets) -> tuple[float, DeviceArray]:
        """Mean silhouette coefficient of points sorted by cluster (cluster c = rows offsets[c]:offsets[c+1])."""
        n, d = x_sorted.shape
        off = np.ascontiguousarray(offsets, np.int64)
        samples = self.empty((n,), np.float64)
        score = self.empty((1,), np.float64)
        check(lib.msm_silhouette(self.handle, x_sorted.ptr, n, d, d, off.ctypes.data, len(off) - 1, samples.ptr, score.ptr),
              self.handle)
        return float(score.to_host()[0]), samples

    # -- regular-grid microstates ---------------------------------------------------------------
    def grid_cells(self, x: DeviceArray, edges: np.ndarray) -> DeviceArray:
        """Flat cell index per frame for edges [F, bins + 1] (np.digitize - 1, clipped)."""
        n, F = x.shape
        e = np.ascontiguousarray(edges, np.float64)
        if e.shape[0] != F:
            raise ValueError("edges must have one row per feature")
        flat = self.empty((n,), np.int32)
        dt = _lib.MSM_F32 if x.dtype == np.float32 else _lib.MSM_F64
        ed = self.to_device(e)
        check(lib.msm_grid_cells(self.handle, x.ptr, dt, n, F, F, ed.ptr, e.shape[1] - 1, flat.ptr), self.handle)
        self.sync()
        return flat

    def first_occurrence(self, flat: DeviceArray, n_cells: int) -> np.ndarray:
        first = self.empty((int(n_cells),), np.int64)
        check(lib.msm_first_occurrence(self.handle, flat.ptr, flat.size, int(n_cells), first.ptr), self.handle)
        return first.to_host()

    def relabel(self, flat: DeviceArray, cell_map: np.ndarray) -> DeviceArray:
        m = self.to_device(np.ascontiguousarray(cell_map, np.int32))
        labels = self.empty((flat.size,), np.int32)
        check(lib.msm_relabel(self.handle, flat.ptr, flat.size, m.ptr, m.size, labels.ptr), self.handle)
        self.sync()
        return labels

    # -- dense solves on T: committors / flux / lumping / MFPT --------------------------------
    def solve(self, A: DeviceArray, B: DeviceArray) -> int:
        """A X = B in place (A -> LU factors, B -> X); returns 0 or the 1-based singular column."""
        n = A.shape[0]
        nrhs = 1 if len(B.shape) == 1 else B.shape[1]
        info = self.empty((1,), np.int32)
        check(lib.msm_solve_f64(self.handle, n, nrhs, A.ptr, A.shape[1], B.ptr, nrhs, info.ptr), self.handle)
        return int(info.to_host()[0])

    def reactive_flux(self, T: DeviceArray, pi: DeviceArray, role: np.ndarray, want_flux: bool = True) -> dict:
        n = T.shape[0]
        rd = self.to_device(np.ascontiguousarray(role, np.int32))
        qp, qm = self.empty((n,), np.float64), self.empty((n,), np.float64)
        info = self.empty((2,), np.int32)
        gross = net = tot = None
        if want_flux:
            gross, net, tot = self.empty((n, n), np.float64), self.empty((n, n), np.float64), self.empty((4,), np.float64)
        check(lib.msm_reactive_flux(self.handle, T.ptr, n, pi.ptr, rd.ptr, n, qp.ptr, qm.ptr,
                                    gross.ptr if gross is not None else None, net.ptr if net is not None else None,
                                    tot.ptr if tot is not None else None, info.ptr), self.handle)
        return {"qplus": qp, "qminus": qm, "gross": gross, "net": net, "totals": tot, "info": info.to_host()}

    def lump_macro(self, T: DeviceArray, pi: DeviceArray, macro: np.ndarray, n_macro: int):
        n = T.shape[0]
        md = self.to_device(np.ascontiguousarray(macro, np.int32))
        Tm, pm = self.empty((n_macro, n_macro), np.float64), self.empty((n_macro,), np.float64)
        check(lib.msm_lump_macro(self.handle, T.ptr, n, pi.ptr, md.ptr, n, int(n_macro), Tm.ptr, pm.ptr), self.handle)
        return Tm, pm

    def macro_mfpt(self, T: DeviceArray):
        n = T.shape[0]
        out = self.empty((n, n), np.float64)
        info = self.empty((n,), np.int32)
        check(lib.msm_macro_mfpt(self.handle, T.ptr, n, n, out.ptr, info.ptr), self.handle)
        return out, info.to_host()

    # -- free-energy surfaces ---------------------------------------------------
    def weighted_stats(self, x: DeviceArray, col: int = 0, weights: DeviceArray | None = None) -> np.ndarray:
        """[sum w, sum w^2, weighted mean, weighted variance, min, max] of column `col` of x [n, d] (or of a
        1-D array); unit weights when `weights` is None."""
        if len(x.shape) == 1:
            n, stride, off = x.shape[0], 1, 0
        else:
            n, stride, off = x.shape[0], x.shape[1], int(col)
        out = self.empty((6,), np.float64)
        check(lib.msm_weighted_stats(self.handle, x.ptr + off * 8, stride, n, weights.ptr if weights is not None else None,
                                     out.ptr), self.handle)
        return out.to_host()

    def gather(self, table: DeviceArray, idx: DeviceArray) -> DeviceArray:
        """table[idx] for int32 indices (0 where the index is out of range)."""
        out = self.empty((idx.size,), np.float64)
        check(lib.msm_gather_f64(self.handle, table.ptr, table.size, idx.ptr, idx.size, out.ptr), self.handle)
        return out

    def hist2d_xy(self, x, y, xedges: np.ndarray, yedges: np.ndarray) -> DeviceArray:
        """np.histogram2d (unweighted) of two device columns on the given edges -> f64 [nx, ny].  x and y are
        (array, column) pairs for columns of 2-D arrays, or 1-D arrays."""
        def col(a):
            if isinstance(a, tuple):
                arr, c = a
                return arr.ptr + int(c) * 8, arr.shape[1], arr.shape[0]
            return a.ptr, 1, a.size
        (px, sx, n), (py, sy, _) = col(x), col(y)
        nx, ny = len(xedges) - 1, len(yedges) - 1
        xe = self.to_device(np.ascontiguousarray(xedges, np.float64))
        ye = self.to_device(np.ascontiguousarray(yedges, np.float64))
        h = self.empty((nx, ny), np.float64)
        check(lib.msm_hist2d(self.handle, px, sx, py, sy, n, None, 0.0, xe.ptr, nx, ye.ptr, ny, h.ptr), self.handle)
        return h

    def clip_or_wrap(self, x: DeviceArray, lo: float, hi: float, *, wrap: bool, col: int = 0) -> DeviceArray:
        """New 1-D array: np.clip(x, lo, hi), or ((x - lo) % (hi - lo)) + lo when wrap."""
        if len(x.shape) == 1:
            n, stride, off = x.shape[0], 1, 0
        else:
            n, stride, off = x.shape[0], x.shape[1], int(col)
        out = self.empty((n,), np.float64)
        check(lib.msm_clip_or_wrap(self.handle, x.ptr + off * 8, stride, n, float(lo), float(hi), 2 if wrap else 1,
                                   out.ptr), self.handle)
        return out

    def order_statistics(self, x: DeviceArray, ranks, col: int = 0) -> np.ndarray:
        """The ranks[q]-th smallest values (0-based) of column `col` of x [n, d] (or of a 1-D array)."""
        if len(x.shape) == 1:
            n, stride, off = x.shape[0], 1, 0
        else:
            n, stride, off = x.shape[0], x.shape[1], int(col)
        r = np.ascontiguousarray(ranks, dtype=np.int64).ravel()
        out = np.empty(r.size, dtype=np.float64)
        check(lib.msm_order_statistics(self.handle, x.ptr + off * 8, stride, n, r.ctypes.data, r.size, out.ctypes.data),
              self.handle)
        return out

    def hist2d(self, x: DeviceArray, cols, xedges: np.ndarray, yedges: np.ndarray,
               weights: DeviceArray | None = None, w_absmax: float = 0.0) -> DeviceArray:
        """np.histogram2d(x[:, cols[0]], x[:, cols[1]], bins=[xedges, yedges], weights=w) -> f64 [nx, ny]."""
        n, d = x.shape
        nx, ny = len(xedges) - 1, len(yedges) - 1
        xe, ye = self.to_device(np.ascontiguousarray(xedges, np.float64)), self.to_device(np.ascontiguousarray(yedges, np.float64))
        h = self.empty((nx, ny), np.float64)
        check(lib.msm_hist2d(self.handle, x.ptr + int(cols[0]) * 8, d, x.ptr + int(cols[1]) * 8, d, n,
                             weights.ptr if weights is not None else None, float(w_absmax), xe.ptr, nx, ye.ptr, ny, h.ptr),
              self.handle)
        return h

    def smooth_sparse_bins(self, hist: DeviceArray, min_count: float) -> tuple[DeviceArray, int]:
        nx, ny = hist.shape
        out = self.empty((nx, ny), np.float64)
        cnt = self.empty((1,), np.int32)
        check(lib.msm_smooth_sparse_bins(self.handle, hist.ptr, nx, ny, float(min_count), out.ptr, cnt.ptr), self.handle)
        return out, int(cnt.to_host()[0])

    def scale_to_total(self, v: DeviceArray, total: float) -> None:
        check(lib.msm_scale_to_total(self.handle, v.ptr, v.size, float(total)), self.handle)

    def fes_finalize(self, hist: DeviceArray, kT: float) -> tuple[DeviceArray, int]:
        F = self.empty(hist.shape, np.float64)
        st = self.empty((1,), np.int32)
        check(lib.msm_fes_finalize(self.handle, hist.ptr, hist.size, float(kT), F.ptr, st.ptr), self.handle)
        return F, int(st.to_host()[0])

    def kde2d(self, x: DeviceArray, cols, xcenters: np.ndarray, ycenters: np.ndarray, bw_x: float, bw_y: float,
              weights: DeviceArray | None, w_scale: float, periodic: int = 0) -> DeviceArray:
        """periodic: bit 0 / bit 1 wrap the x / y differences to [-pi, pi) (density on a torus)."""
        n, d = x.shape
        xc, yc = self.to_device(np.ascontiguousarray(xcenters, np.float64)), self.to_device(np.ascontiguousarray(ycenters, np.float64))
        dens = self.empty((len(xcenters), len(ycenters)), np.float64)
        check(lib.msm_kde2d(self.handle, x.ptr + int(cols[0]) * 8, d, x.ptr + int(cols[1]) * 8, d, n,
                            weights.ptr if weights is not None else None, float(w_scale), xc.ptr, len(xcenters), yc.ptr,
                            len(ycenters), float(bw_x), float(bw_y), int(periodic), dens.ptr), self.handle)
        return dens

    def gemm(self, A: DeviceArray, B: DeviceArray, out: DeviceArray | None = None) -> DeviceArray:
        """C = A . B (fp64, matrix cores; ascending-k FMA chain per element)."""
        m, k = A.shape
        k2, n = B.shape
        if k != k2:
            raise ValueError(f"gemm: inner dimensions differ ({k} vs {k2})")
        C = out if out is not None else self.empty((m, n), np.float64)
        check(lib.msm_gemm_f64(self.handle, m, n, k, A.ptr, k, B.ptr, n, C.ptr, n), self.handle)
        return C

    def ck_test(self, T1: DeviceArray, Tk: DeviceArray, factors, rowcounts: DeviceArray | None = None):
        """mse[i] = mean((T1^f_i - Tk[i])^2) and, with row counts [F, n], the multinomial noise RMS
        of Tk[i] (validation/ck_rule.py:36-63).  T1 [n, n]; Tk [F, n, n]."""
        n = T1.shape[0]
        fac = np.ascontiguousarray(factors, dtype=np.int32)
        F = int(fac.size)
        if tuple(Tk.shape) != (F, n, n):
            raise ValueError(f"ck_test: Tk must have shape ({F}, {n}, {n}), got {tuple(Tk.shape)}")
        mse = self.empty((max(F, 1),), np.float64)
        noise = self.empty((max(F, 1),), np.float64) if rowcounts is not None else None
        check(lib.msm_ck_test(self.handle, T1.ptr, n, Tk.ptr, n * n, n, n, fac.ctypes.data, F,
                              rowcounts.ptr if rowcounts is not None else None, n, mse.ptr,
                              noise.ptr if noise is not None else None), self.handle)
        return mse.to_host()[:F], (noise.to_host()[:F] if noise is not None else None)

    def matrix_power(self, T: DeviceArray, squarings: int, *, n: DeviceArray | None = None) -> DeviceArray:
        """T^(2^squarings) of one matrix [k,k] or a batch [B,k,k] (orders in `n`), msm_matrix_power."""
        B = T.shape[0] if len(T.shape) == 3 else 1
        k = T.shape[-1]
        out = self.empty(T.shape, np.float64)
        tmp = self.empty(T.shape, np.float64) if squarings > 1 else None
        check(lib.msm_matrix_power(self.handle, T.ptr, k * k, k, n.ptr if n is not None else None, k, B, int(squarings),
                                   tmp.ptr if tmp is not None else None, out.ptr), self.handle)
        return out

    def spectrum(self, T: DeviceArray, *, n: DeviceArray | None = None, n_its: int = 0, lags=None,
                 p: int | None = None, want_pi: bool = True, tol: float = 1e-9, n_iter: int = 24,
                 max_launches: int = 100, seed: int = 0, allow_unconverged: bool = False, n_vecs: int = 0,
                 n_watch: int | None = None, squarings: int | None = None) -> dict:
        """Leading Ritz values / stationary distribution / implied timescales of one matrix
        [k,k] or a batch [B,k,k] of packed row-stochastic matrices (orders in `n`, int32 [B]).
        squarings: the subspace iterations run on T^(2^squarings) (default 2 for k >= 32: a quarter of the iterations,
        the reported values are those of T itself); when a run on the power does not converge it is repeated on T."""
        if squarings is None:
            squarings = int(os.environ.get("MSM_SPEC_SQUARINGS", "2")) if T.shape[-1] >= 32 else 0
        if squarings > 0:
            # One matrix: a ramp of 6, 6, 12, 24 iterations per launch (a well-separated spectrum is done after the first
            # six on T^4: 0.69 ms at k = 200, 0.89 ms at k = 500; four sufficed at k = 500 but not at k = 200); batches: two thirds of the usual number (converged matrices are frozen between
            # launches only).  Any matrix left unconverged (or a basis T^(2^s) has made too ill-conditioned to
            # factorise: NaN) sends the whole call to the plain iteration on T below.
            B = T.shape[0] if len(T.shape) == 3 else 1
            it_pow = int(os.environ.get("MSM_SPEC_NITER", -n_iter if B == 1 else max(6, (2 * n_iter) // 3)))
            try:
                return self._spectrum(T, self.matrix_power(T, squarings, n=n), it_pow, n=n, n_its=n_its, lags=lags, p=p,
                                      want_pi=want_pi, tol=tol, max_launches=max(8, max_launches // 2), seed=seed,
                                      allow_unconverged=False, n_vecs=n_vecs, n_watch=n_watch)
            except _lib.MsmError:
                pass
        return self._spectrum(T, None, n_iter, n=n, n_its=n_its, lags=lags, p=p, want_pi=want_pi, tol=tol,
                              max_launches=max_launches, seed=seed, allow_unconverged=allow_unconverged, n_vecs=n_vecs,
                              n_watch=n_watch)

    def _spectrum(self, T: DeviceArray, Tpow: DeviceArray | None, n_iter: int, *, n, n_its, lags, p, want_pi, tol,
                  max_launches, seed, allow_unconverged, n_vecs, n_watch) -> dict:
        batched = len(T.shape) == 3
        B = T.shape[0] if batched else 1
        k = T.shape[-1]
        n_vecs = int(min(n_vecs, 32, k))
        watch = int(n_watch) if n_watch is not None else max(n_its + 1, n_vecs, 1)
        if p is None:
            p = min(32, max(watch + 6, 8))
        p = int(min(p, 32, k))
        ws_bytes = int(lib.msm_spectrum_workspace_bytes(k, p, B))
        # the solver's buffers are kept between calls of the same shape (a lag scan calls this once
        # per estimate; nine hipMallocs cost more than the solve of a small matrix); pi is returned
        # to the caller and therefore always fresh
        cache = getattr(self, "_spec_cache", None)
        key = (k, B, max(n_its, 1))
        m_its = max(n_its, 1)
        if cache is None or cache[0] != key or cache[1].size < ws_bytes:
            # the small outputs share ONE allocation [ritz B x 128 | change B | its_eig B x m | its_ts B x m | status B i32]:
            # one copy to the host at the end instead of five
            outs = self.empty((B * (129 + 2 * m_its) + (B + 1) // 2,), np.float64)
            cache = (key, self.empty((ws_bytes,), np.uint8), outs)
            self._spec_cache = cache
        _, ws, outs = cache
        o_change, o_eig, o_ts, o_status = B * 128, B * 129, B * (129 + m_its), B * (129 + 2 * m_its)
        ritz = self.wrap(outs.ptr, (B, 128), np.float64)
        change = self.wrap(outs.ptr + 8 * o_change, (B,), np.float64)
        its_eig = self.wrap(outs.ptr + 8 * o_eig, (B, m_its), np.float64)
        its_ts = self.wrap(outs.ptr + 8 * o_ts, (B, m_its), np.float64)
        status = self.wrap(outs.ptr + 8 * o_status, (B,), np.int32)
        pi = self.empty((B, k), np.float64) if want_pi else None
        vecs = self.empty((B, n_vecs, k), np.float64) if n_vecs else None
        lag_d = self.to_device(np.asarray(lags if lags is not None else np.ones(B), np.float64).reshape(B))
        launches = 0
        restart = True
        since_restart = 0
        worst = prev = float("inf")
        ramp = n_iter < 0          # negative: ramp up to |n_iter|
        n_iter_max = abs(int(n_iter))
        for launch in range(max_launches):
            n_iter = min(n_iter_max, 6 << max(0, launch - 1)) if ramp else n_iter_max
            tail = (k * k, k, n.ptr if n is not None else None, k, B, p, int(n_iter), int(restart), int(seed), watch, ws.ptr,
                    ritz.ptr, pi.ptr if pi is not None else None, k, change.ptr, status.ptr, int(n_its), lag_d.ptr,
                    its_eig.ptr, its_ts.ptr, float(tol) if B > 1 else 0.0, vecs.ptr if vecs is not None else None, n_vecs)
            if Tpow is not None:
                check(lib.msm_spectrum_powered(self.handle, T.ptr, Tpow.ptr, *tail), self.handle)
            else:
                check(lib.msm_spectrum(self.handle, T.ptr, *tail), self.handle)
            launches += 1
            restart = False
            since_restart += 1
            prev, worst = worst, float(np.max(change.to_host()))
            if os.environ.get("MSM_SPEC_TRACE") == "2":
                print(f"  launch {launches}: worst residual {worst:.3e}", file=sys.stderr)
            if worst <= tol:
                break
            if Tpow is not None and not np.isfinite(worst):
                raise _lib.MsmError("msm_spectrum: the basis broke down under the powered iteration")
            # slow convergence = the watched eigenvalues are close to lambda_{p+1}: a wider basis moves
            # that ratio down at little cost per iteration.  The decay of the worst residual between two
            # launches predicts how many more this basis needs; more than a handful -> restart with the
            # widest subspace right away (converged matrices of a batch are frozen, not re-iterated)
            if since_restart >= 2 and p < min(32, k):
                rate = worst / prev if 0.0 < prev < float("inf") else 1.0
                remaining = np.log(tol / worst) / np.log(rate) if 0.0 < rate < 1.0 else float("inf")
                if remaining > 6:
                    p = int(min(32, k))
                    ws = self.empty((int(lib.msm_spectrum_workspace_bytes(k, p, B)),), np.uint8)
                    self._spec_cache = (key, ws, outs)
                    restart, since_restart, worst = True, 0, float("inf")
        else:
            if not allow_unconverged:
                raise _lib.MsmError(
                    f"msm_spectrum: residual {worst:.3e} > {tol:.1e} after {launches} launches "
                    "(leading eigenvalues too clustered for subspace iteration)")
        host = outs.to_host()
        ch = host[o_change:o_change + B].copy()
        if os.environ.get("MSM_SPEC_TRACE"):
            print(f"msm_spectrum: B={B} k={k} p={p} powered={Tpow is not None} n_iter={n_iter} launches={launches} "
                  f"worst={worst:.2e} unconverged={int(np.sum(ch > tol))}", file=sys.stderr)
        st = host[o_status:].view(np.int32)[:B]
        if np.any(st != 0):
            raise _lib.MsmError(f"msm_spectrum: hqr did not converge (status {st.tolist()})")
        r = host[:B * 128].reshape(B, 128)
        out = {"ritz": r[:, :32] + 1j * r[:, 32:64], "p": p, "launches": launches, "residual": ch, "pi": pi, "vecs": vecs}
        if n_its:
            out["its_eig"] = host[o_eig:o_eig + B * m_its].reshape(B, m_its).copy()
            out["its_ts"] = host[o_ts:o_ts + B * m_its].reshape(B, m_its).copy()
        return out


    def sample_transition_matrices(self, counts: DeviceArray, active: DeviceArray, n_active: DeviceArray, *,
                                   alpha: float, seed: int, n_samples: int, first_sample: int = 0,
                                   out: DeviceArray | None = None) -> DeviceArray:
        """n_samples posterior draws [n_samples, k, k] (packed active blocks) of a mode-1 estimate."""
        k = counts.shape[-1]
        if out is None:
            out = self.empty((int(n_samples), k, k), np.float64)
        done = 0
        while done < n_samples:                       # grid.y limit: 65535 samples per launch
            m = min(65535, n_samples - done)
            check(lib.msm_sample_transition_matrices(
                self.handle, counts.ptr, int(counts.dtype == np.float64), k, active.ptr, n_active.ptr, float(alpha),
                int(seed) & (2 ** 64 - 1), int(first_sample + done), m, out.ptr + done * k * k * 8, k * k, k),
                self.handle)
            done += m
        return out

    def reversible_mle(self, counts: DeviceArray, *, maxerr: float = 1e-8, maxiter: int = 1_000_000) -> dict:
        """Reversible maximum-likelihood T and pi of a connected f64 count matrix [n, n]."""
        import ctypes

        n = counts.shape[0]
        T, pi = self.empty((n, n), np.float64), self.empty((n,), np.float64)
        it, err = ctypes.c_int(0), ctypes.c_double(0.0)
        check(lib.msm_reversible_mle(self.handle, counts.ptr, n, counts.shape[1], float(maxerr), int(maxiter), T.ptr, n,
                                     pi.ptr, ctypes.byref(it), ctypes.byref(err)), self.handle)
        return {"T": T, "pi": pi, "iterations": it.value, "err": err.value}

    def diff_norms(self, P: DeviceArray, Q: DeviceArray) -> np.ndarray:
        """[sum |P - Q|, sum |Q|, sum (P - Q)^2] of two f64 matrices of one shape."""
        n, m = P.shape
        out = self.empty((3,), np.float64)
        check(lib.msm_diff_norms(self.handle, P.ptr, m, Q.ptr, Q.shape[1], n, m, out.ptr), self.handle)
        return out.to_host()

    def philox4x32(self, key: int, counter) -> np.ndarray:
        import ctypes

        out = self.empty((4,), np.uint32)
        c = (ctypes.c_uint32 * 4)(*[int(v) & 0xFFFFFFFF for v in counter])
        check(lib.msm_philox4x32(self.handle, int(key) & (2 ** 64 - 1), c, out.ptr), self.handle)
        return out.to_host()


_ENGINES: dict[int, Engine] = {}


def get_engine(device: int = 0) -> Engine:
    """Process-wide engine per device (operator shims share it)."""
    eng = _ENGINES.get(device)
    if eng is None or not eng.handle:
        eng = Engine(device)
        _ENGINES[device] = eng
    return eng
