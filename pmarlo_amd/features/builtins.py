"""Built-in featurizers on the GPU: mirror of pmarlo.features.builtins
(S/features/builtins.py:42-86 PhiPsiFeature, :281-395 Distance/Angle/DihedralFeature).

The objects keep the reference's stateful protocol (``labels`` / periodic flags set as a
side effect of ``compute``), and ``astype(float)`` outputs with NaN -> 0 (:307,346,387)."""

from __future__ import annotations

import logging

import numpy as np

from ..device import get_engine
from .base import register_feature

__all__ = ["PhiPsiFeature", "Chi1Feature", "DistanceFeature", "AngleFeature", "DihedralFeature",
           "RadiusOfGyrationFeature", "DistancePairFeature", "ContactsPairFeature", "SASAFeature",
           "HBondsCountFeature", "SecondaryStructureFractionFeature"]


def _device_features(traj, **kw) -> np.ndarray:
    eng = get_engine()
    out = eng.featurize(eng.to_device(np.ascontiguousarray(traj.xyz, np.float32)), **kw)
    return out.to_host()


class PhiPsiFeature:
    name = "phi_psi"

    def __init__(self) -> None:
        self._periodic: np.ndarray | None = None
        self.labels: list[str] | None = None

    def compute(self, traj, **kwargs) -> np.ndarray:
        phi_idx = traj.topology.phi_indices()
        psi_idx = traj.topology.psi_indices()
        if len(phi_idx) == 0 and len(psi_idx) == 0:
            self.labels = []
            return np.zeros((traj.n_frames, 0), dtype=float)
        quads = np.vstack([phi_idx, psi_idx])
        X = _device_features(traj, quads=quads)  # wrapped to (-pi, pi] by the kernel
        self.labels = ([f"phi:res{int(traj.topology.res_index[q[1]])}" for q in phi_idx]
                       + [f"psi:res{int(traj.topology.res_index[q[2]])}" for q in psi_idx])
        self._periodic = np.ones((X.shape[1],), dtype=bool)
        return X

    def is_periodic(self) -> np.ndarray:
        return np.empty((0,), dtype=bool) if self._periodic is None else self._periodic


class Chi1Feature:
    """Side-chain chi1 of every residue with a gamma atom, wrapped to (-pi, pi], periodic
    (S/features/builtins.py:138-168; mdtraj.compute_chi1's atom patterns)."""

    name = "chi1"

    def __init__(self) -> None:
        self._periodic: np.ndarray | None = None
        self.labels: list[str] | None = None

    def compute(self, traj, **kwargs) -> np.ndarray:
        quads = traj.topology.chi1_indices()
        if len(quads) == 0:
            self.labels = []
            return np.zeros((traj.n_frames, 0), dtype=float)
        X = _device_features(traj, quads=quads)
        self._periodic = np.ones((X.shape[1],), dtype=bool)
        self.labels = [f"chi1:res{int(traj.topology.res_index[q[1]])}" for q in quads]
        return X

    def is_periodic(self) -> np.ndarray:
        return np.empty((0,), dtype=bool) if self._periodic is None else self._periodic


class _IndexedFeature:
    width = 0
    kw = ""
    periodic = False

    def __init__(self) -> None:
        self._periodic: np.ndarray | None = None
        self.labels: list[str] | None = None

    def compute(self, traj, indices=None, **kwargs) -> np.ndarray:
        if indices is None or len(indices) != self.width:
            raise ValueError(f"{self.name} requires {self.width} atom indices")
        idx = [int(i) for i in indices]
        if min(idx) < 0 or max(idx) >= traj.n_atoms:
            raise ValueError(f"{self.name} indices out of range")
        X = _device_features(traj, **{self.kw: [idx]}).astype(float)
        X = np.nan_to_num(X, nan=0.0)
        self._periodic = np.array([self.periodic], dtype=bool)
        self.labels = [f"{self.name}:" + "-".join(str(i) for i in idx)]
        return X

    def is_periodic(self) -> np.ndarray:
        return np.empty((0,), dtype=bool) if self._periodic is None else self._periodic


class DistanceFeature(_IndexedFeature):
    name, width, kw, periodic = "distance", 2, "pairs", False


class AngleFeature(_IndexedFeature):
    name, width, kw, periodic = "angle", 3, "triplets", False


class DihedralFeature(_IndexedFeature):
    name, width, kw, periodic = "dihedral", 4, "quads", True


class RadiusOfGyrationFeature:
    """``Rg``: radius of gyration with unit masses (S/features/builtins.py:89-105)."""

    name = "Rg"

    def __init__(self) -> None:
        self._periodic = np.array([False], dtype=bool)
        self.labels: list[str] | None = None

    def compute(self, traj, **kwargs) -> np.ndarray:
        eng = get_engine()
        out = eng.featurize_rg(eng.to_device(np.ascontiguousarray(traj.xyz, np.float32)))
        self.labels = ["Rg"]
        return out.to_host().reshape(-1, 1).astype(float)

    def is_periodic(self) -> np.ndarray:
        return self._periodic


class _PairKwFeature:
    """Features addressed by ``i=, j=`` keywords (S/features/builtins.py:109-135, 252-275)."""

    def __init__(self) -> None:
        self._periodic = np.array([False], dtype=bool)
        self.labels: list[str] | None = None

    @staticmethod
    def _pair(traj, kwargs) -> list[int]:
        i, j = int(kwargs.get("i", -1)), int(kwargs.get("j", -1))
        if not (0 <= i < traj.n_atoms) or not (0 <= j < traj.n_atoms):
            raise ValueError("Atom indices out of range")
        return [i, j]

    def is_periodic(self) -> np.ndarray:
        return self._periodic


class DistancePairFeature(_PairKwFeature):
    name = "distance_pair"

    def compute(self, traj, **kwargs) -> np.ndarray:
        i, j = self._pair(traj, kwargs)
        d = _device_features(traj, pairs=[[i, j]]).astype(float)
        self.labels = [f"dist:atoms:{i}-{j}"]
        return np.nan_to_num(d, nan=0.0, posinf=0.0, neginf=0.0)


class ContactsPairFeature(_PairKwFeature):
    name = "contacts_pair"

    def compute(self, traj, **kwargs) -> np.ndarray:
        rcut = float(kwargs.get("rcut", 0.5))
        if rcut <= 0:
            raise ValueError("rcut must be positive")
        i, j = self._pair(traj, kwargs)
        eng = get_engine()
        out = eng.featurize_contacts(eng.to_device(np.ascontiguousarray(traj.xyz, np.float32)), [[i, j]], rcut)
        return out.to_host().astype(float)


def _zeros_after_failure(feature: str, exc: Exception, shape: tuple[int, int]) -> np.ndarray:
    """The reference wraps the mdtraj call of ``sasa`` / ``hbonds_count`` / ``ssfrac`` in ``except Exception`` and
    continues with a column of zeros (S/features/builtins.py:176-246); so does the engine, for failures of the
    COMPUTATION (an element without a radius, a device-side limit such as more neighbours than the kernel's list
    holds), and says so in the log.  A missing library or device is not such a failure: there is no CPU path to
    fall back to, so that error travels on."""
    if isinstance(exc, ImportError):
        raise exc
    logging.getLogger("pmarlo").warning("feature %r failed (%s: %s); the reference's fallback applies: zeros", feature,
                                        type(exc).__name__, exc)
    return np.zeros(shape, dtype=float)


class SASAFeature:
    """``sasa``: total Shrake-Rupley solvent accessible surface per frame, the sum of the per-residue areas
    (S/features/builtins.py:171-188).  Like the reference, any failure of the computation (an element without a
    radius, ...) yields a column of zeros."""

    name = "sasa"

    def __init__(self) -> None:
        self._periodic = np.array([False], dtype=bool)
        self.labels: list[str] | None = None

    def compute(self, traj, **kwargs) -> np.ndarray:
        self.labels = ["sasa"]
        get_engine()   # no device, no library: that error is not one the zero column stands for
        try:
            from .structure import shrake_rupley

            sasa = shrake_rupley(traj, mode="residue")  # (n_frames, n_residues)
            return np.sum(sasa, axis=1, keepdims=True).astype(float)
        except Exception as exc:
            return _zeros_after_failure("sasa", exc, (traj.n_frames, 1))

    def is_periodic(self) -> np.ndarray:
        return self._periodic


class HBondsCountFeature:
    """``hbonds_count``: the number of Baker-Hubbard hydrogen bonds of the TRAJECTORY (present in more than 10 % of
    its frames), repeated for every frame -- the reference's "static count" (S/features/builtins.py:194-213)."""

    name = "hbonds_count"

    def __init__(self) -> None:
        self._periodic = np.array([False], dtype=bool)
        self.labels: list[str] | None = None

    def compute(self, traj, **kwargs) -> np.ndarray:
        self.labels = ["hbonds_count"]
        get_engine()   # no device, no library: that error is not one the zero column stands for
        try:
            from .structure import baker_hubbard

            hbonds = baker_hubbard(traj, periodic=True)
            return np.full((traj.n_frames, 1), float(len(hbonds)), dtype=float)
        except Exception as exc:
            return _zeros_after_failure("hbonds_count", exc, (traj.n_frames, 1))

    def is_periodic(self) -> np.ndarray:
        return self._periodic


class SecondaryStructureFractionFeature:
    """``ssfrac``: fractions of helix (H, G, I), sheet (E, B) and coil residues per frame from the DSSP codes
    (S/features/builtins.py:219-246); 'NA' residues count in the denominator, as in the reference."""

    name = "ssfrac"

    def __init__(self) -> None:
        self._periodic = np.array([False, False, False], dtype=bool)
        self.labels: list[str] | None = None

    def compute(self, traj, **kwargs) -> np.ndarray:
        self.labels = ["ssfrac:helix", "ssfrac:sheet", "ssfrac:coil"]
        get_engine()   # no device, no library: that error is not one the zero column stands for
        try:
            from .structure import compute_dssp

            dssp = compute_dssp(traj)  # (n_frames, n_residues) of 'H' / 'E' / 'C' / 'NA'
            n = float(dssp.shape[1]) if dssp.shape[1] > 0 else 1.0
            # the reference counts per row with float division, then coil = 1 - helix - sheet clipped at 0
            helix = np.isin(dssp, ["H", "G", "I"]).sum(axis=1) / n
            sheet = np.isin(dssp, ["E", "B"]).sum(axis=1) / n
            coil = np.maximum(0.0, 1.0 - helix - sheet)
            return np.stack([helix, sheet, coil], axis=1).astype(float)
        except Exception as exc:
            return _zeros_after_failure("ssfrac", exc, (traj.n_frames, 3))

    def is_periodic(self) -> np.ndarray:
        return self._periodic


for _cls in (PhiPsiFeature, Chi1Feature, DistanceFeature, AngleFeature, DihedralFeature, RadiusOfGyrationFeature,
             DistancePairFeature, ContactsPairFeature, SASAFeature, HBondsCountFeature,
             SecondaryStructureFractionFeature):
    register_feature(_cls())
