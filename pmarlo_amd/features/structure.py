"""Solvent accessible surface, hydrogen bonds and secondary structure on the device: what the reference's built-ins
``sasa``, ``hbonds_count`` and ``ssfrac`` ask mdtraj for (S/features/builtins.py:171-250: ``md.shrake_rupley(traj,
mode="residue")``, ``md.baker_hubbard(traj, periodic=True)``, ``md.compute_dssp(traj)``).

mdtraj is not part of this build; the three functions restate its published algorithms (mdtraj 1.10:
``geometry/sasa.py`` + ``src/sasa.cpp``, ``geometry/hbond.py``, ``geometry/dssp.py`` + ``src/dssp.cpp``) with the
same names, keyword defaults and return shapes, the per-frame arithmetic on the GPU (``csrc/structure.hip``).
Parity with mdtraj itself is therefore unpinned; ``oracle/npport.py`` holds the CPU restatement the kernels are
checked against.

Bonds.  mdtraj takes N-H / O-H bonds from its residue templates; a PDB read here has no bond table, so a hydrogen
is bonded to the nearest heavy atom of its own residue in the first frame (within 0.13 nm), which is the same
chemistry for any sane structure."""

from __future__ import annotations

import numpy as np

from ..device import get_engine
from ..io.selection import PROTEIN_RESIDUES, WATER_RESIDUES

__all__ = ["ATOMIC_RADII", "sphere_points", "shrake_rupley", "hbond_triplets", "baker_hubbard", "compute_dssp",
           "backbone_table"]

# van der Waals radii in nm by element (mdtraj.geometry.sasa._ATOMIC_RADII, Bondi / Mantina et al.)
ATOMIC_RADII = {
    "H": 0.120, "He": 0.140, "Li": 0.076, "Be": 0.059, "B": 0.192, "C": 0.170, "N": 0.155, "O": 0.152, "F": 0.147,
    "Ne": 0.154, "Na": 0.102, "Mg": 0.086, "Al": 0.184, "Si": 0.210, "P": 0.180, "S": 0.180, "Cl": 0.181,
    "Ar": 0.188, "K": 0.138, "Ca": 0.114, "Sc": 0.211, "Ti": 0.200, "V": 0.200, "Cr": 0.200, "Mn": 0.200,
    "Fe": 0.200, "Co": 0.200, "Ni": 0.163, "Cu": 0.140, "Zn": 0.139, "Ga": 0.187, "Ge": 0.211, "As": 0.185,
    "Se": 0.190, "Br": 0.185, "Kr": 0.202, "I": 0.198, "Xe": 0.216,
}


def sphere_points(n_points: int) -> np.ndarray:
    """Golden-section spiral on the unit sphere (sasa.cpp generate_sphere_points): float32 [n_points, 3]."""
    i = np.arange(n_points, dtype=np.float64)
    inc = np.pi * (3.0 - np.sqrt(5.0))
    offset = 2.0 / n_points
    y = i * offset - 1.0 + offset / 2.0
    r = np.sqrt(1.0 - y * y)
    phi = i * inc
    return np.stack([np.cos(phi) * r, y, np.sin(phi) * r], axis=1).astype(np.float32)


def _elements(top) -> list[str]:
    return [str(e).capitalize() for e in top.elements]


def shrake_rupley(traj, probe_radius: float = 0.14, n_sphere_points: int = 960, mode: str = "atom",
                  change_radii: dict | None = None) -> np.ndarray:
    """``mdtraj.shrake_rupley``: float32 (n_frames, n_atoms) or, ``mode="residue"``, (n_frames, n_residues) in nm^2."""
    top = traj.topology
    radii_tab = dict(ATOMIC_RADII)
    if change_radii:
        radii_tab.update(change_radii)
    try:
        radii = np.asarray([radii_tab[e] for e in _elements(top)], dtype=np.float32)
    except KeyError as exc:
        raise ValueError(f"no van der Waals radius for element {exc.args[0]!r}") from exc
    radii = radii + np.float32(probe_radius)
    if mode not in ("atom", "residue"):
        raise ValueError('mode must be one of "residue", "atom". "%s" supplied' % mode)
    eng = get_engine()
    xyz = eng.to_device(np.ascontiguousarray(traj.xyz, np.float32))
    areas = eng.featurize_sasa(xyz, radii, sphere_points(int(n_sphere_points))).to_host()
    if mode == "atom":
        return areas
    return residue_sums(areas, np.asarray(top.res_index), top.n_residues)


def residue_sums(areas: np.ndarray, res_index: np.ndarray, n_residues: int) -> np.ndarray:
    """Per-residue sums in atom order, float32 (sasa.cpp: ``out[atom_mapping[i]] += areas[i]``)."""
    out = np.zeros((areas.shape[0], n_residues), dtype=np.float32)
    order = np.argsort(res_index, kind="stable")
    ranks = np.zeros(len(res_index), dtype=int)          # position of each atom inside its residue
    seen: dict[int, int] = {}
    for a in order:
        r = int(res_index[a])
        ranks[a] = seen.get(r, 0)
        seen[r] = ranks[a] + 1
    for p in range(int(ranks.max()) + 1 if len(ranks) else 0):
        sel = np.nonzero(ranks == p)[0]
        out[:, res_index[sel]] += areas[:, sel]
    return out


def _is_water(top) -> np.ndarray:
    return np.isin(np.asarray(top.res_names), list(WATER_RESIDUES))


def hbond_triplets(traj, exclude_water: bool = True, sidechain_only: bool = False) -> np.ndarray:
    """(donor, hydrogen, acceptor) index triplets of ``hbond._get_bond_triplets``: donors are N-H and O-H bonds,
    acceptors every N and O atom, water left out on request, donor != acceptor."""
    top = traj.topology
    el = np.asarray(_elements(top))
    water = _is_water(top)
    res = np.asarray(top.res_index)
    xyz0 = np.asarray(traj.xyz[0], dtype=np.float64)
    names = np.asarray(top.atom_names)
    backbone = np.isin(names, ["N", "CA", "C", "O", "H", "HA"])
    donors = []
    for h in np.nonzero(el == "H")[0]:
        if exclude_water and water[h]:
            continue
        mates = np.nonzero((res == res[h]) & (el != "H"))[0]
        if len(mates) == 0:
            continue
        dist = np.linalg.norm(xyz0[mates] - xyz0[h], axis=1)
        j = int(mates[np.argmin(dist)])
        if dist.min() < 0.13 and el[j] in ("N", "O"):
            if sidechain_only and (backbone[j] or backbone[h]):
                continue
            donors.append((j, int(h)))
    # mdtraj lists the N-H donors before the O-H donors
    donors = [d for d in donors if el[d[0]] == "N"] + [d for d in donors if el[d[0]] == "O"]
    acc_mask = np.isin(el, ["O", "N"])
    if exclude_water:
        acc_mask &= ~water
    if sidechain_only:
        acc_mask &= ~backbone
    acceptors = np.nonzero(acc_mask)[0]
    if not donors or len(acceptors) == 0:
        return np.zeros((0, 3), dtype=int)
    dh = np.asarray(donors, dtype=int)
    trip = np.empty((len(dh) * len(acceptors), 3), dtype=int)
    trip[:, 0] = np.repeat(dh[:, 0], len(acceptors))
    trip[:, 1] = np.repeat(dh[:, 1], len(acceptors))
    trip[:, 2] = np.tile(acceptors, len(dh))
    return trip[trip[:, 0] != trip[:, 2]]


def baker_hubbard(traj, freq: float = 0.1, exclude_water: bool = True, periodic: bool = True,
                  sidechain_only: bool = False, distance_cutoff: float = 0.25,
                  angle_cutoff: float = 2.0 * np.pi / 3.0) -> np.ndarray:
    """``mdtraj.baker_hubbard``: the (donor, hydrogen, acceptor) triplets whose H...A distance is below 0.25 nm and
    whose D-H...A angle exceeds 120 degrees in more than ``freq`` of the frames.  ``periodic`` has no effect on
    coordinates without a unit cell, as in mdtraj."""
    trip = hbond_triplets(traj, exclude_water=exclude_water, sidechain_only=sidechain_only)
    if len(trip) == 0 or traj.n_frames == 0:
        return np.zeros((0, 3), dtype=int)
    eng = get_engine()
    xyz = eng.to_device(np.ascontiguousarray(traj.xyz, np.float32))
    counts = eng.hbond_presence(xyz, trip, distance_cutoff, np.float32(angle_cutoff))
    return trip[counts / float(traj.n_frames) > freq]


def backbone_table(top):
    """Protein residues with N, CA, C and O (what dssp.py hands to the C code): their residue indices, the atom
    index table int32 [R, 4], chain index per residue and the proline flags."""
    res = np.asarray(top.res_index)
    names = top.atom_names
    n_res = top.n_residues
    table = -np.ones((n_res, 4), dtype=np.int32)
    col = {"N": 0, "CA": 1, "C": 2, "O": 3}
    for a, nm in enumerate(names):
        c = col.get(nm)
        if c is not None and table[res[a], c] < 0:
            table[res[a], c] = a
    first_atom = np.full(n_res, -1, dtype=int)
    for a in range(len(names) - 1, -1, -1):
        first_atom[res[a]] = a
    resn = [top.res_names[a] if a >= 0 else "" for a in first_atom]
    is_protein = np.asarray([nm in PROTEIN_RESIDUES for nm in resn])
    keep = np.nonzero(is_protein & (table >= 0).all(axis=1))[0]
    chain = np.asarray([int(top.chain_index[first_atom[r]]) for r in keep], dtype=np.int32)
    proline = np.asarray([resn[r] == "PRO" for r in keep], dtype=np.uint8)
    return keep, table[keep], chain, proline


_FULL = np.array([" ", "H", "B", "E", "G", "I", "T", "S"])
_SIMPLE = np.array(["C", "H", "E", "E", "H", "H", "C", "C"])


def compute_dssp(traj, simplified: bool = True) -> np.ndarray:
    """``mdtraj.compute_dssp``: (n_frames, n_residues) array of codes; simplified 'H' (H, G, I), 'E' (E, B), 'C',
    full ' ', 'H', 'B', 'E', 'G', 'I', 'T', 'S'; residues that are not protein or lack a backbone atom get 'NA'."""
    top = traj.topology
    keep, table, chain, proline = backbone_table(top)
    out = np.full((traj.n_frames, top.n_residues), "NA", dtype="<U2")
    if len(keep) == 0 or traj.n_frames == 0:
        return out
    eng = get_engine()
    xyz = eng.to_device(np.ascontiguousarray(traj.xyz, np.float32))
    codes = eng.dssp(xyz, table, chain, proline)
    out[:, keep] = (_SIMPLE if simplified else _FULL)[codes]
    return out
