"""featurize_trajectory: mirror of pmarlo.features.featurize (S/features/featurize.py:17-66)
with the geometry on the GPU, plus trig_expand_periodic (S/api/features.py:138-180)."""

from __future__ import annotations

import numpy as np

from ..device import get_engine

__all__ = ["featurize_trajectory", "trig_expand_periodic"]

_SUPPORTED = ("phi_psi", "ca_distances", "backbone_torsions")


def featurize_trajectory(traj, feature_type: str = "phi_psi") -> np.ndarray:
    """(n_frames, n_features) float32: ``"phi_psi"`` -> [phi..., psi...] radians;
    ``"ca_distances"`` -> all i<j C-alpha pairs (nm); ``"backbone_torsions"`` -> phi, psi and chi1
    (N-CA-CB-gamma of every residue that has a gamma atom, residue order)."""
    eng = get_engine()
    if feature_type in ("phi_psi", "backbone_torsions"):
        blocks = [traj.topology.phi_indices(), traj.topology.psi_indices()]
        if feature_type == "backbone_torsions":
            blocks.append(traj.topology.chi1_indices())
        quads = np.vstack(blocks)
        xyz = eng.to_device(np.ascontiguousarray(traj.xyz, np.float32))
        return eng.featurize(xyz, quads=quads).to_host()
    if feature_type == "ca_distances":
        ca = traj.topology.select("name CA")
        if len(ca) < 2:
            raise ValueError("Topology has fewer than 2 Cα atoms.")
        pairs = np.array([(ca[i], ca[j]) for i in range(len(ca)) for j in range(i + 1, len(ca))], dtype=np.int32)
        xyz = eng.to_device(np.ascontiguousarray(traj.xyz, np.float32))
        return eng.featurize(xyz, pairs=pairs).to_host()
    raise ValueError(f"Unknown feature_type {feature_type!r}. Choose one of {_SUPPORTED}.")


def trig_expand_periodic(X: np.ndarray, periodic: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Periodic column j -> adjacent [cos, sin]; mapping[k] = source column of Xe[:, k].
    Elementwise host transform of an already-materialised matrix; the fused device form
    is ``Engine.featurize(..., dihedral_mode=1)``."""
    X = np.asarray(X)
    periodic = np.asarray(periodic)
    if X.size == 0:
        return X, np.array([], dtype=int)
    if periodic.size != X.shape[1]:
        raise ValueError(
            f"periodic array size ({periodic.size}) must match number of features ({X.shape[1]})")
    cols, mapping = [], []
    for j in range(X.shape[1]):
        if bool(periodic[j]):
            cols += [np.cos(X[:, j]), np.sin(X[:, j])]
            mapping += [j, j]
        else:
            cols.append(X[:, j])
            mapping.append(j)
    return np.vstack(cols).T, np.asarray(mapping, dtype=int)
