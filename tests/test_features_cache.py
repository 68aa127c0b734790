"""Feature cache key / file format (pmarlo_amd.api.features vs S/api/features.py:27-107).
The key is pinned by a known answer assembled by hand from the documented recipe."""
import hashlib
import json

import numpy as np
import pytest

from pmarlo_amd.api.features import compute_features, feature_cache_file
from pmarlo_amd.io import Topology, Trajectory


def _toy():
    names = ["N", "CA", "C", "O", "N", "CA", "C", "O"]
    top = Topology(names, ["ALA"] * 4 + ["GLY"] * 4, np.array([0, 0, 0, 0, 1, 1, 1, 1]), ["A"] * 8)
    xyz = (np.arange(25 * 8 * 3, dtype=np.float32).reshape(25, 8, 3) * 0.0137).astype(np.float32)
    return Trajectory(xyz, top)


def test_cache_key_known_answer(tmp_path):
    traj = _toy()
    specs = ["phi_psi", "distance([0, 5])"]
    top_hash = hashlib.sha1(json.dumps([8, 2, 1, traj.topology.atom_names, ["ALA", "GLY"]],
                                       separators=(",", ":")).encode()).hexdigest()
    sample = traj.xyz[::2, :8, :]                       # 25 frames, nf = 10 -> step 2; 8 < 50 atoms
    pos_hash = hashlib.sha1((sample * 1000.0).round().astype("int32").tobytes()).hexdigest()
    meta = {"n_frames": 25, "n_atoms": 8, "specs": specs, "top_hash": top_hash, "pos_hash": pos_hash}
    key = hashlib.sha1(json.dumps(meta, sort_keys=True, separators=(",", ":")).encode()).hexdigest()
    got = feature_cache_file(traj, specs, str(tmp_path / "cache"))
    assert got == tmp_path / "cache" / f"features_{key}.npz"
    assert (tmp_path / "cache").is_dir()
    assert feature_cache_file(traj, specs, None) is None
    # any change of coordinates, specs or topology changes the key
    other = Trajectory(traj.xyz + np.float32(0.01), traj.topology)
    assert feature_cache_file(other, specs, str(tmp_path)) != feature_cache_file(traj, specs, str(tmp_path))
    assert feature_cache_file(traj, specs[:1], str(tmp_path)) != feature_cache_file(traj, specs, str(tmp_path))


def test_cached_entry_is_returned_verbatim(tmp_path):
    traj = _toy()
    specs = ["phi_psi"]
    f = feature_cache_file(traj, specs, str(tmp_path))
    X = np.arange(50, dtype=float).reshape(25, 2)
    np.savez_compressed(f, X=X, columns=np.array(["phi_0", "psi_0"], dtype=np.str_), periodic=np.array([True, True]))
    got, cols, per = compute_features(traj, specs, cache_path=str(tmp_path))   # no GPU touched: cache hit
    np.testing.assert_array_equal(got, X)
    assert cols == ["phi_0", "psi_0"] and per.tolist() == [True, True]


@pytest.mark.gpu
def test_compute_features_writes_and_reloads_cache(tmp_path):
    traj = _toy()
    specs = ["phi_psi", "distance([0, 5])", "dihedral([0, 1, 2, 4])"]
    X, cols, per = compute_features(traj, specs, cache_path=str(tmp_path))
    assert X.shape == (25, 4) and len(cols) == 4 and per.tolist() == [True, True, False, True]
    f = feature_cache_file(traj, specs, str(tmp_path))
    assert f.exists()
    with np.load(f) as data:
        np.testing.assert_array_equal(data["X"], X)
        assert data["columns"].astype(str).tolist() == cols
    X2, cols2, per2 = compute_features(traj, specs, cache_path=str(tmp_path))
    np.testing.assert_array_equal(X2, X)
    assert cols2 == cols and per2.tolist() == per.tolist()
