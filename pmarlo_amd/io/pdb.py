"""PDB reader and the slice of the mdtraj Trajectory/Topology interface that the
featurizer needs (the input side of the path).

The reference hands ``mdtraj.Trajectory`` objects to its featurizers
(S/features/featurize.py:17-66, S/io/trajectory.py:136-177).  mdtraj's C readers are
out of scope; this module gives the same attributes for PDB input: ``xyz`` float32
(n_frames, n_atoms, 3) in nm, ``n_frames``, ``n_atoms``, ``topology.select("name CA")``,
and the backbone dihedral index tables of ``mdtraj.compute_phi / compute_psi``."""

from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

__all__ = ["Topology", "Trajectory", "load_pdb"]


def _guess_element(atom_name: str) -> str:
    """Element symbol from a PDB atom name when columns 77-78 are blank: leading digits dropped, two-letter symbols
    only for the common ions / halogens."""
    nm = atom_name.strip().lstrip("0123456789")
    if not nm:
        return ""
    two = nm[:2].upper()
    if two in ("CL", "BR", "NA", "MG", "ZN", "FE", "MN", "CU", "LI") and len(nm) == 2:
        return two[0] + two[1].lower()
    return nm[0].upper()


@dataclass
class Topology:
    atom_names: list[str]
    res_names: list[str]
    res_index: np.ndarray          # per-atom 0-based residue index (in file order)
    chain_ids: list[str] = field(default_factory=list)
    res_seq: np.ndarray | None = None      # per-atom PDB residue number (mdtraj resSeq); default res_index + 1
    elements: list[str] | None = None      # per-atom element symbol; default: derived from the atom name
    serials: np.ndarray | None = None      # per-atom PDB serial; default index + 1

    def __post_init__(self):
        n = len(self.atom_names)
        self.res_index = np.asarray(self.res_index, dtype=int)
        if self.res_seq is None:
            self.res_seq = self.res_index + 1
        if self.serials is None:
            self.serials = np.arange(1, n + 1)
        if self.elements is None:
            self.elements = [_guess_element(nm) for nm in self.atom_names]
        # 0-based chain index in order of first appearance (mdtraj chainid)
        seen: dict[str, int] = {}
        self.chain_index = np.asarray([seen.setdefault(c, len(seen)) for c in (self.chain_ids or [""] * n)], dtype=int)

    @property
    def n_atoms(self) -> int:
        return len(self.atom_names)

    @property
    def n_residues(self) -> int:
        return int(self.res_index.max()) + 1 if len(self.res_index) else 0

    def select(self, query: str) -> np.ndarray:
        """mdtraj-style selection strings (pmarlo_amd/io/selection.py: name / resname / resid / resSeq / index /
        chainid / element, ranges, comparisons, protein / backbone / sidechain / water, and / or / not)."""
        from .selection import select as _select

        return _select(self, query)

    def _atom(self, res: int, name: str) -> int | None:
        for i in np.nonzero(self.res_index == res)[0]:
            if self.atom_names[i] == name:
                return int(i)
        return None

    def _same_chain(self, r1: int, r2: int) -> bool:
        if not self.chain_ids:
            return True
        a = np.nonzero(self.res_index == r1)[0]
        b = np.nonzero(self.res_index == r2)[0]
        return len(a) > 0 and len(b) > 0 and self.chain_ids[a[0]] == self.chain_ids[b[0]]

    def phi_indices(self) -> np.ndarray:
        """C(i-1), N(i), CA(i), C(i) for every residue that has them (mdtraj.compute_phi)."""
        out = []
        for r in range(1, self.n_residues):
            if not self._same_chain(r - 1, r):
                continue
            quad = [self._atom(r - 1, "C"), self._atom(r, "N"), self._atom(r, "CA"), self._atom(r, "C")]
            if None not in quad:
                out.append(quad)
        return np.asarray(out, dtype=np.int32).reshape(-1, 4)

    def psi_indices(self) -> np.ndarray:
        """N(i), CA(i), C(i), N(i+1) (mdtraj.compute_psi)."""
        out = []
        for r in range(0, self.n_residues - 1):
            if not self._same_chain(r, r + 1):
                continue
            quad = [self._atom(r, "N"), self._atom(r, "CA"), self._atom(r, "C"), self._atom(r + 1, "N")]
            if None not in quad:
                out.append(quad)
        return np.asarray(out, dtype=np.int32).reshape(-1, 4)

    def chi1_indices(self) -> np.ndarray:
        """N, CA, CB and the gamma atom (CG, CG1, SG, OG or OG1) of every residue that has them, in
        residue order (mdtraj.compute_chi1: one pattern matches per residue type)."""
        out = []
        for r in range(self.n_residues):
            head = [self._atom(r, "N"), self._atom(r, "CA"), self._atom(r, "CB")]
            if None in head:
                continue
            for gamma in ("CG", "CG1", "SG", "OG", "OG1"):
                g = self._atom(r, gamma)
                if g is not None:
                    out.append(head + [g])
                    break
        return np.asarray(out, dtype=np.int32).reshape(-1, 4)


@dataclass
class Trajectory:
    xyz: np.ndarray                # float32 (n_frames, n_atoms, 3), nanometres
    topology: Topology

    def __post_init__(self):
        self.xyz = np.ascontiguousarray(self.xyz, dtype=np.float32)
        if self.xyz.ndim != 3 or self.xyz.shape[2] != 3 or self.xyz.shape[1] != self.topology.n_atoms:
            raise ValueError(f"xyz shape {self.xyz.shape} does not match topology ({self.topology.n_atoms} atoms)")

    @property
    def n_frames(self) -> int:
        return int(self.xyz.shape[0])

    @property
    def n_atoms(self) -> int:
        return int(self.xyz.shape[1])

    def __len__(self) -> int:
        return self.n_frames

    def __getitem__(self, key) -> "Trajectory":
        xyz = self.xyz[key]
        if xyz.ndim == 2:
            xyz = xyz[None]
        return Trajectory(xyz, self.topology)


def load_pdb(path: str | Path) -> Trajectory:
    """Read ATOM/HETATM records of every MODEL (coordinates in Angstrom -> nm)."""
    models: list[list[list[float]]] = []
    cur: list[list[float]] = []
    names: list[str] = []
    resn: list[str] = []
    chains: list[str] = []
    res_keys: list[tuple] = []
    serials: list[int] = []
    resseq: list[int] = []
    elems: list[str] = []
    first = True
    with open(path) as fh:
        for line in fh:
            rec = line[:6]
            if rec in ("ATOM  ", "HETATM"):
                cur.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
                if first:
                    names.append(line[12:16].strip())
                    resn.append(line[17:20].strip())
                    chains.append(line[21:22])
                    res_keys.append((line[21:22], line[22:27]))
                    try:
                        serials.append(int(line[6:11]))
                    except ValueError:
                        serials.append(len(serials) + 1)
                    try:
                        resseq.append(int(line[22:26]))
                    except ValueError:
                        resseq.append(0)
                    el = line[76:78].strip() if len(line) >= 78 else ""
                    elems.append(el.capitalize() if el else _guess_element(line[12:16]))
            elif rec == "ENDMDL":
                if cur:
                    models.append(cur)
                cur = []
                first = False
    if cur:
        models.append(cur)
    if not models:
        raise ValueError(f"{path}: no ATOM records")
    n_atoms = len(names)
    if any(len(m) != n_atoms for m in models):
        raise ValueError(f"{path}: models differ in atom count")
    res_index = np.zeros(n_atoms, dtype=int)
    idx, last = -1, None
    for i, key in enumerate(res_keys):
        if key != last:
            idx += 1
            last = key
        res_index[i] = idx
    xyz = np.asarray(models, dtype=np.float64) / 10.0
    return Trajectory(xyz.astype(np.float32), Topology(names, resn, res_index, chains, np.asarray(resseq), elems,
                                                      np.asarray(serials)))
