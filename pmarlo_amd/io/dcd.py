"""DCD trajectory reader / writer (the input side of the path).

The reference streams DCD files through ``mdtraj.iterload`` (S/io/trajectory.py:136-177,
S/markov_state_model/_loading.py:21-228) and hands ``Trajectory`` chunks to the featurizers.
mdtraj is not a dependency of this engine; this module restates the published CHARMM/NAMD DCD
layout on numpy memory maps, so a shard is never read whole into host memory:

  record 1  int32 84 | 'CORD' | int32 icntrl[20] | int32 84
            icntrl[0] n_frames, [1] first step, [2] step interval, [8] fixed atoms,
            [9] time step (float32), [10] 1 = a unit-cell record precedes every frame,
            [11] 1 = a fourth coordinate record follows every frame, [19] CHARMM version
  record 2  int32 size | int32 n_title | n_title x 80 chars | int32 size
  record 3  int32 4 | int32 n_atoms | int32 4
  frame     [int32 48 | 6 float64 (a, gamma, b, beta, alpha, c) | int32 48]
            int32 4N | N float32 x | int32 4N ; same for y ; same for z ; [same for w]

Coordinates are Angstrom on disk and nanometres in memory (as mdtraj delivers them).  Either
byte order is accepted (detected from the first record marker).  Files with fixed atoms
(icntrl[8] != 0) store partial frames and are refused.
"""

from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Iterator, Sequence

import numpy as np

from .pdb import Topology, Trajectory

__all__ = ["DCDFile", "load_dcd", "iterload", "write_dcd"]


@dataclass
class _Layout:
    order: str            # '<' or '>'
    n_frames: int
    n_atoms: int
    has_cell: bool
    has_4d: bool
    first_frame: int      # byte offset of frame 0
    frame_bytes: int
    istart: int
    nsavc: int
    delta: float
    titles: list[str]


def _read_layout(path: Path) -> _Layout:
    size = path.stat().st_size
    with open(path, "rb") as fh:
        head = fh.read(92)
        if len(head) < 92:
            raise ValueError(f"{path}: too short for a DCD header")
        order = None
        for cand in ("<", ">"):
            if int(np.frombuffer(head[:4], dtype=cand + "i4")[0]) == 84:
                order = cand
        if order is None or head[4:8] != b"CORD":
            raise ValueError(f"{path}: not a DCD file (no 84/'CORD' header record)")
        icntrl = np.frombuffer(head[8:88], dtype=order + "i4")
        if int(np.frombuffer(head[88:92], dtype=order + "i4")[0]) != 84:
            raise ValueError(f"{path}: corrupt DCD header record")
        if int(icntrl[8]) != 0:
            raise NotImplementedError(f"{path}: DCD files with fixed atoms are not supported")
        delta = float(np.frombuffer(head[8 + 36:8 + 40], dtype=order + "f4")[0])
        i4 = np.dtype(order + "i4")
        rec = int(np.frombuffer(fh.read(4), dtype=i4)[0])
        body = fh.read(rec)
        if len(body) != rec or int(np.frombuffer(fh.read(4), dtype=i4)[0]) != rec:
            raise ValueError(f"{path}: corrupt DCD title record")
        n_title = int(np.frombuffer(body[:4], dtype=i4)[0]) if rec >= 4 else 0
        titles = [body[4 + 80 * t:4 + 80 * (t + 1)].decode("ascii", "replace").rstrip("\x00 ") for t in range(n_title)
                  if 4 + 80 * (t + 1) <= rec]
        blk = np.frombuffer(fh.read(12), dtype=i4)
        if blk.size != 3 or int(blk[0]) != 4 or int(blk[2]) != 4:
            raise ValueError(f"{path}: corrupt DCD atom-count record")
        n_atoms = int(blk[1])
        first = fh.tell()
    has_cell = int(icntrl[10]) == 1
    has_4d = int(icntrl[11]) == 1
    frame_bytes = (56 if has_cell else 0) + (4 if has_4d else 3) * (8 + 4 * n_atoms)
    n_frames = int(icntrl[0])
    avail = (size - first) // frame_bytes if frame_bytes else 0
    if n_frames <= 0 or n_frames > avail:   # writers that never patched the header, or truncated files
        n_frames = int(avail)
    return _Layout(order, n_frames, n_atoms, has_cell, has_4d, first, frame_bytes, int(icntrl[1]), int(icntrl[2]),
                   delta, titles)


class DCDFile:
    """Random access to the frames of one DCD file through a read-only memory map."""

    def __init__(self, path: str | Path):
        self.path = Path(path)
        self.layout = _read_layout(self.path)
        lay = self.layout
        self._mm = np.memmap(self.path, dtype=np.uint8, mode="r") if lay.n_frames else None

    @property
    def n_frames(self) -> int:
        return self.layout.n_frames

    @property
    def n_atoms(self) -> int:
        return self.layout.n_atoms

    def __len__(self) -> int:
        return self.n_frames

    def read(self, start: int = 0, stop: int | None = None, stride: int = 1,
             atom_indices: Sequence[int] | None = None) -> tuple[np.ndarray, np.ndarray | None]:
        """(xyz float32 (n, n_sel, 3) in nm, cell (n, 6) [a, b, c in nm, alpha, beta, gamma] or None)."""
        lay = self.layout
        stop = lay.n_frames if stop is None else min(int(stop), lay.n_frames)
        start = max(0, int(start))
        stride = max(1, int(stride))
        frames = range(start, stop, stride)
        sel = None if atom_indices is None else np.asarray(atom_indices, dtype=np.int64)
        if sel is not None and sel.size and (sel.min() < 0 or sel.max() >= lay.n_atoms):
            raise ValueError("atom_indices out of range")
        n_sel = lay.n_atoms if sel is None else int(sel.size)
        xyz = np.empty((len(frames), n_sel, 3), dtype=np.float32)
        cell = np.empty((len(frames), 6), dtype=np.float64) if lay.has_cell else None
        f4, f8, i4 = np.dtype(lay.order + "f4"), np.dtype(lay.order + "f8"), np.dtype(lay.order + "i4")
        rec = 8 + 4 * lay.n_atoms
        for out_i, fr in enumerate(frames):
            off = lay.first_frame + fr * lay.frame_bytes
            if lay.has_cell:
                raw = np.frombuffer(self._mm, dtype=f8, count=6, offset=off + 4)
                # on disk: a, gamma, b, beta, alpha, c (angles as degrees, or cosines in old CHARMM files)
                a, gam, b, bet, alp, c = (float(v) for v in raw)
                ang = np.array([alp, bet, gam])
                if np.all(np.abs(ang) <= 1.0):  # cosines
                    ang = np.degrees(np.arccos(ang))
                cell[out_i] = [a / 10.0, b / 10.0, c / 10.0, ang[0], ang[1], ang[2]]
                off += 56
            for ax in range(3):
                base = off + ax * rec
                if int(np.frombuffer(self._mm, dtype=i4, count=1, offset=base)[0]) != 4 * lay.n_atoms:
                    raise ValueError(f"{self.path}: corrupt coordinate record in frame {fr}")
                col = np.frombuffer(self._mm, dtype=f4, count=lay.n_atoms, offset=base + 4)
                xyz[out_i, :, ax] = col if sel is None else col[sel]
        xyz *= np.float32(0.1)
        return xyz, cell


def _as_topology(top, n_atoms: int, atom_indices) -> Topology:
    if isinstance(top, Trajectory):
        top = top.topology
    if top is None:
        names = [f"A{i}" for i in range(n_atoms)]
        top = Topology(names, ["UNK"] * n_atoms, np.zeros(n_atoms, dtype=int), [" "] * n_atoms)
    if top.n_atoms != n_atoms:
        raise ValueError(f"topology has {top.n_atoms} atoms, the DCD file {n_atoms}")
    if atom_indices is None:
        return top
    sel = np.asarray(atom_indices, dtype=int)
    _, res = np.unique(top.res_index[sel], return_inverse=True)
    return Topology([top.atom_names[i] for i in sel], [top.res_names[i] for i in sel], res,
                    [top.chain_ids[i] for i in sel] if top.chain_ids else [])


def load_dcd(filename: str | Path, top=None, stride: int = 1, atom_indices: Sequence[int] | None = None) -> Trajectory:
    """Whole file (strided / atom-sliced) as one Trajectory; `top` = Topology, Trajectory (e.g. from
    load_pdb) or None."""
    f = DCDFile(filename)
    xyz, _ = f.read(0, None, stride, atom_indices)
    return Trajectory(xyz, _as_topology(top, f.n_atoms, atom_indices))


def iterload(filename: str | Path, *, top=None, stride: int = 1, atom_indices: Sequence[int] | None = None,
             chunk: int = 1000) -> Iterator[Trajectory]:
    """Stream a DCD file in chunks of `chunk` (strided) frames: the signature of
    pmarlo.io.trajectory.iterload (S/io/trajectory.py:136-177)."""
    f = DCDFile(filename)
    topo = _as_topology(top, f.n_atoms, atom_indices)
    stride = max(1, int(stride))
    chunk = max(1, int(chunk))
    span = chunk * stride
    for start in range(0, f.n_frames, span):
        xyz, _ = f.read(start, min(f.n_frames, start + span), stride, atom_indices)
        if xyz.shape[0]:
            yield Trajectory(xyz, topo)


def write_dcd(filename: str | Path, xyz_nm: np.ndarray, cell: np.ndarray | None = None, *, big_endian: bool = False,
              title: str = "written by pmarlo_amd") -> None:
    """Write (n_frames, n_atoms, 3) nm coordinates as a CHARMM-format DCD (tests, shard export)."""
    xyz = np.asarray(xyz_nm, dtype=np.float64) * 10.0
    if xyz.ndim != 3 or xyz.shape[2] != 3:
        raise ValueError("xyz must be (n_frames, n_atoms, 3)")
    n_frames, n_atoms = xyz.shape[:2]
    o = ">" if big_endian else "<"
    i4 = np.dtype(o + "i4")
    icntrl = np.zeros(20, dtype=i4)
    icntrl[0], icntrl[1], icntrl[2], icntrl[3] = n_frames, 0, 1, n_frames
    icntrl[10] = 1 if cell is not None else 0
    icntrl[19] = 24
    head = bytearray(icntrl.tobytes())
    head[36:40] = np.asarray([1.0], dtype=o + "f4").tobytes()
    with open(filename, "wb") as fh:
        fh.write(np.asarray([84], dtype=i4).tobytes() + b"CORD" + bytes(head) + np.asarray([84], dtype=i4).tobytes())
        t = title.encode("ascii", "replace")[:80].ljust(80)
        fh.write(np.asarray([84], dtype=i4).tobytes() + np.asarray([1], dtype=i4).tobytes() + t
                 + np.asarray([84], dtype=i4).tobytes())
        fh.write(np.asarray([4, n_atoms, 4], dtype=i4).tobytes())
        mark = np.asarray([4 * n_atoms], dtype=i4).tobytes()
        for fr in range(n_frames):
            if cell is not None:
                a, b, c, alp, bet, gam = (float(v) for v in np.asarray(cell)[fr])
                fh.write(np.asarray([48], dtype=i4).tobytes()
                         + np.asarray([a * 10.0, gam, b * 10.0, bet, alp, c * 10.0], dtype=o + "f8").tobytes()
                         + np.asarray([48], dtype=i4).tobytes())
            for ax in range(3):
                fh.write(mark + xyz[fr, :, ax].astype(o + "f4").tobytes() + mark)
