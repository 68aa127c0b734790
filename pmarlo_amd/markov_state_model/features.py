"""Feature computation of the MSM object: mirror of FeaturesMixin.compute_features
(S/markov_state_model/_features.py:23-97 driver, :131-142 phi/psi block layout
[cos phi | sin phi | cos psi | sin psi], :144-171 phi/psi + distances and the C-alpha distance
selection -- every third C-alpha, j >= i + 3, capped at ``n_features or 200`` pairs, :181-231 the
optional TICA step) with the geometry and the TICA on the GPU.

Not mirrored: the ``universal*`` metric (VAMP/PCA reducers) and ``contacts`` (mdtraj residue
contact scheme)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from ..device import get_engine
from .reduction import tica_fit_transform_trajectories

__all__ = ["MSMFeatures", "compute_msm_features", "ca_distance_pairs"]


@dataclass
class MSMFeatures:
    features: np.ndarray          # stacked over trajectories (after the optional TICA step)
    raw_frames: int
    strided_frames: int
    effective_frames: int
    traj_lengths: List[int]       # frames per trajectory in `features`


def ca_distance_pairs(ca_indices: Sequence[int], n_features: Optional[int]) -> np.ndarray:
    """The reference's C-alpha pair list (_features.py:155-170)."""
    ca = [int(i) for i in ca_indices]
    if len(ca) < 2:
        raise ValueError("Insufficient Cα atoms for distance features")
    n_pairs = min(n_features or 200, len(ca) * (len(ca) - 1) // 2)
    pairs: list[list[int]] = []
    for i in range(0, len(ca), 3):
        for j in range(i + 3, len(ca), 3):
            pairs.append([ca[i], ca[j]])
            if len(pairs) >= n_pairs:
                return np.asarray(pairs, dtype=np.int32)
    return np.asarray(pairs, dtype=np.int32).reshape(-1, 2)


def _phi_psi_block(eng, xd, traj) -> np.ndarray:
    phi, psi = traj.topology.phi_indices(), traj.topology.psi_indices()
    blocks = [eng.featurize(xd, quads=q, dihedral_mode=2).to_host() for q in (phi, psi) if len(q)]
    if not blocks:   # no backbone dihedrals: the reference substitutes one period of a sine / cosine ramp
        t = np.linspace(0.0, 1.0, traj.n_frames, endpoint=False, dtype=np.float32)
        return np.column_stack([np.sin(2.0 * np.pi * t), np.cos(2.0 * np.pi * t)])
    return np.hstack(blocks)


def _features_for_traj(eng, traj, feature_type: str, n_features: Optional[int]) -> np.ndarray:
    ft = feature_type.lower()
    xd = eng.to_device(np.ascontiguousarray(traj.xyz, np.float32))
    if ft.startswith("phi_psi_distances"):
        pairs = ca_distance_pairs(traj.topology.select("name CA"), n_features)
        return np.hstack([_phi_psi_block(eng, xd, traj), eng.featurize(xd, pairs=pairs).to_host()])
    if ft.startswith("phi_psi"):
        return _phi_psi_block(eng, xd, traj)
    if ft == "distances":
        return eng.featurize(xd, pairs=ca_distance_pairs(traj.topology.select("name CA"), n_features)).to_host()
    if ft.startswith("universal") or ft == "contacts":
        raise NotImplementedError(f"feature type {feature_type!r} is outside the accelerated path")
    raise ValueError(f"Unknown feature type: {feature_type}")


def compute_msm_features(trajectories: Sequence, feature_type: str = "phi_psi", n_features: Optional[int] = None,
                         feature_stride: int = 1, tica_lag: int = 0, tica_components: Optional[int] = None) -> MSMFeatures:
    """Stride every trajectory, featurize it, stack; then (when ``tica_components`` is given, ``tica_lag``
    > 0 or the type name contains "tica") project with TICA: dimensions clamped to [2, 5] and the last
    ``tica_lag`` frames of every trajectory dropped, as the reference does."""
    stride = int(max(1, feature_stride))
    lag = int(max(0, tica_lag))
    raw = sum(int(t.n_frames) for t in trajectories)
    eng = get_engine()
    strided = [t[::stride] for t in trajectories]
    blocks = [_features_for_traj(eng, t, feature_type, n_features) for t in strided]
    lengths = [int(b.shape[0]) for b in blocks]
    X = np.vstack(blocks) if blocks else np.empty((0, 0))
    if tica_components is not None or lag > 0 or "tica" in feature_type.lower():
        hint = tica_components or n_features
        if hint is not None:                  # _maybe_apply_tica returns early without a dimension hint
            X, _ = tica_fit_transform_trajectories(X, lengths, int(hint), lag)
            lengths = [n - lag if n > lag else 0 for n in lengths]
    return MSMFeatures(X, raw, sum(int(t.n_frames) for t in strided), int(X.shape[0]), lengths)
