"""compute_features with the on-disk feature cache: mirror of pmarlo.api.features
(S/api/features.py:27-107 cache file, :192-208 compute_features, :319-342 block assembly).

The cache is the only on-disk format on the path.  File name and
payload follow the reference, so caches written by either implementation are picked up by the
other: ``<cache_path>/features_<sha1>.npz`` holding ``X``, ``columns`` (str) and ``periodic``;
the key is the sha1 of the compact, key-sorted JSON of

    n_frames, n_atoms, specs,
    top_hash = sha1(json [n_atoms, n_residues, n_chains, first 50 atom names, first 50 residue names]),
    pos_hash = sha1 of round(1000 * xyz[::max(1, n_frames // min(n_frames, 10)), :50]) as int32.
"""

from __future__ import annotations

import hashlib
import json
import logging
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ..features import get_feature, parse_feature_spec
from ..features.featurize import trig_expand_periodic

__all__ = ["compute_features", "feature_cache_file", "trig_expand_periodic"]

logger = logging.getLogger("pmarlo")


def _compact(obj, **kw) -> bytes:
    return json.dumps(obj, separators=(",", ":"), **kw).encode()


def _residue_names(top) -> list[str]:
    """One name per residue, in file order (what iterating mdtraj's ``top.residues`` gives)."""
    if hasattr(top, "residues") and not hasattr(top, "res_index"):
        return [r.name for r in top.residues]
    names: list[str] = []
    last = None
    for i, r in enumerate(np.asarray(top.res_index)):
        if r != last:
            names.append(top.res_names[i])
            last = r
    return names


def _atom_names(top) -> list[str]:
    return list(top.atom_names) if hasattr(top, "atom_names") else [a.name for a in top.atoms]


def _n_chains(top) -> int:
    if hasattr(top, "chains") and not hasattr(top, "chain_ids"):
        return len(list(top.chains))
    ids = list(getattr(top, "chain_ids", []) or [])
    if not ids:
        return 1
    n, last = 0, None
    for c in ids:
        if c != last:
            n += 1
            last = c
    return n


def feature_cache_file(traj, feature_specs: Sequence[str], cache_path: Optional[str]) -> Optional[Path]:
    """Path of the cache entry for (trajectory, specs), or None when caching is off.  Creates the
    cache directory.  Same key as the reference's _resolve_cache_file (S/api/features.py:27-77)."""
    if not cache_path:
        return None
    folder = Path(cache_path)
    folder.mkdir(parents=True, exist_ok=True)
    top = traj.topology
    atoms, residues = _atom_names(top), _residue_names(top)
    top_hash = hashlib.sha1(_compact([len(atoms), len(residues), _n_chains(top), atoms[:50], residues[:50]])).hexdigest()
    pos_hash = None
    xyz = traj.xyz
    if xyz is not None and np.size(xyz):
        n_frames = int(traj.n_frames)
        nf = min(n_frames, 10)
        na = min(int(traj.n_atoms), 50)
        sample = np.asarray(xyz[::max(1, n_frames // nf), :na, :], dtype=np.float32)
        pos_hash = hashlib.sha1((sample * 1000.0).round().astype(np.int32).tobytes()).hexdigest()
    meta = {"n_frames": int(traj.n_frames), "n_atoms": int(traj.n_atoms), "specs": list(feature_specs),
            "top_hash": top_hash, "pos_hash": pos_hash}
    return folder / f"features_{hashlib.sha1(_compact(meta, sort_keys=True)).hexdigest()}.npz"


def _column_labels(fc, name: str, n_cols: int, kwargs: dict) -> List[str]:
    labels = getattr(fc, "labels", None)
    if isinstance(labels, list) and len(labels) == n_cols:
        return list(labels)
    if name == "phi_psi" and n_cols > 0:
        half = n_cols // 2
        return [f"phi_{i}" for i in range(half)] + [f"psi_{i}" for i in range(n_cols - half)]
    base = name
    if name == "distance_pair" and "i" in kwargs and "j" in kwargs:
        base = f"dist:atoms:{kwargs['i']}-{kwargs['j']}"
    return [base] if n_cols == 1 else [f"{base}_{i}" for i in range(n_cols)]


def _compute_blocks(traj, feature_specs: Sequence[str]) -> Tuple[np.ndarray, List[str], np.ndarray]:
    blocks: list[np.ndarray] = []
    flags: list[np.ndarray] = []
    columns: List[str] = []
    for spec in feature_specs:
        name, kwargs = parse_feature_spec(spec)
        fc = get_feature(name)
        X = fc.compute(traj, **kwargs)
        logger.info("[features] %-14s -> shape=%s", name, tuple(X.shape))
        if X.size == 0:
            continue
        periodic = np.asarray(fc.is_periodic())
        if name in {"distance", "distance_pair"} and bool(periodic.any()):
            raise ValueError(f"Distance features must not be flagged as periodic; feature '{name}' returned "
                             f"periodicity {periodic}")
        blocks.append(X)
        flags.append(periodic)
        columns.extend(_column_labels(fc, name, X.shape[1], kwargs))
    if not blocks:
        return np.empty((traj.n_frames, 0), dtype=float), columns, np.empty((0,), dtype=bool)
    lengths = [int(b.shape[0]) for b in blocks]
    n_min = min(lengths)
    if any(n != n_min for n in lengths):
        logger.warning("[features] Frame count mismatch across features: %s -> truncating to %d", lengths, n_min)
    X_all = np.hstack([b[:n_min] for b in blocks])
    return X_all, columns, np.concatenate(flags) if flags else np.zeros((X_all.shape[1],), dtype=bool)


def compute_features(traj, feature_specs: Sequence[str], cache_path: Optional[str] = None
                     ) -> Tuple[np.ndarray, List[str], np.ndarray]:
    """(X, columns, periodic) for the feature specs, e.g. ``["phi_psi", "distance([0, 5])"]``.
    With ``cache_path`` the result is loaded from / saved to ``features_<sha1>.npz`` there."""
    cache_file = feature_cache_file(traj, feature_specs, cache_path)
    if cache_file is not None and cache_file.exists():
        with np.load(cache_file) as data:
            X, cols, periodic = data["X"], [str(c) for c in data["columns"].tolist()], data["periodic"]
        logger.info("[features] Loaded from cache %s: shape=%s, columns=%d", cache_file, tuple(X.shape), len(cols))
        return X, cols, periodic
    X, columns, periodic = _compute_blocks(traj, feature_specs)
    if cache_file is not None:
        np.savez_compressed(cache_file, X=X, columns=np.array(columns, dtype=np.str_), periodic=periodic)
    return X, columns, periodic
