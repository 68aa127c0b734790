"""HIP featurizer kernels vs the golden outputs of the reference's torch extractor and
the numpy restatement."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import npport

pytestmark = pytest.mark.gpu


def test_chignolin_ca_distances_golden(engine, golden):
    g = golden("featurizer.npz")
    out = engine.featurize(engine.to_device(g["chig_xyz"]), pairs=g["chig_pairs"]).to_host()
    assert out.shape == (36, 45) and out.dtype == np.float32
    np.testing.assert_allclose(out, g["chig_dist"], rtol=2e-6)


def test_alanine_phi_psi_and_angles_golden(engine, golden):
    g = golden("featurizer.npz")
    xyz = engine.to_device(g["ala_xyz"])
    out = engine.featurize(xyz, triplets=g["ala_triplets"], quads=g["ala_quads"]).to_host()
    np.testing.assert_allclose(out[:, :3], g["ala_angles"], atol=2e-5)
    np.testing.assert_allclose(out[:, 3:], g["ala_dihedrals"], atol=2e-5)
    # layouts: interleaved (api/features.py trig_expand) and block (_features.py phi/psi)
    inter = engine.featurize(xyz, quads=g["ala_quads"], dihedral_mode=1).to_host()
    block = engine.featurize(xyz, quads=g["ala_quads"], dihedral_mode=2).to_host()
    phi_psi = g["ala_dihedrals"].astype(np.float64)
    want_inter, mapping = npport.trig_expand_periodic(phi_psi, np.array([True, True]))
    np.testing.assert_array_equal(mapping, [0, 0, 1, 1])
    np.testing.assert_allclose(inter, want_inter, atol=3e-5)
    np.testing.assert_allclose(block, np.hstack([np.cos(phi_psi), np.sin(phi_psi)]), atol=3e-5)


def test_unit_cube_known_answer(engine, golden):
    """tests/features/deeptica/test_ts_feature_extractor.py:44-74."""
    g = golden("featurizer.npz")
    out = engine.featurize(engine.to_device(g["cube_xyz"]), pairs=[[0, 1]], triplets=[[0, 1, 2]],
                           quads=[[0, 1, 2, 3]]).to_host()
    np.testing.assert_allclose(out[0], [1.0, np.pi / 2, np.pi / 2], atol=1e-6)
    np.testing.assert_allclose(out[0], g["cube_feats"], atol=1e-6)


def test_large_random_vs_numpy_and_degenerate(engine):
    rng = np.random.default_rng(0)
    n, A = 200_000, 22
    xyz = rng.normal(size=(n, A, 3)).astype(np.float32)
    xyz[5, 3] = xyz[5, 2]                 # zero-length bond -> clamped, no NaN
    pairs = [(i, j) for i in range(6) for j in range(i + 1, 6)]
    quads = [[4, 6, 8, 14], [6, 8, 14, 16], [2, 3, 4, 5]]
    out = engine.featurize(engine.to_device(xyz), pairs=pairs, quads=quads).to_host()
    assert np.isfinite(out).all()
    np.testing.assert_allclose(out[:, :15], npport.distances(xyz, pairs), rtol=3e-6, atol=1e-6)
    ref = npport.dihedrals(xyz, quads)
    diff = np.abs(out[:, 15:] - ref)
    diff = np.minimum(diff, 2 * np.pi - diff)     # +-pi branch
    assert np.percentile(diff, 99.9) < 1e-4 and diff.max() < 5e-2
    assert out[:, 15:].max() <= np.float32(np.pi) and out[:, 15:].min() > -np.float32(np.pi)


def test_bad_indices_raise(engine):
    xyz = engine.to_device(np.zeros((2, 4, 3), np.float32))
    with pytest.raises(ValueError):
        engine.featurize(xyz, pairs=[[0, 4]])


def test_rg_distance_pair_contacts_pair(engine, golden):
    """S/features/builtins.py:89-135, 252-275 (mdtraj.compute_rg with unit masses; contact = d <= rcut)."""
    from pmarlo_amd.api.features import compute_features
    from pmarlo_amd.features import get_feature
    from pmarlo_amd.io import Topology, Trajectory

    g = golden("featurizer.npz")
    xyz = np.tile(g["chig_xyz"], (40, 1, 1)).astype(np.float32)
    xyz += np.random.default_rng(0).normal(0, 0.03, size=xyz.shape).astype(np.float32)
    A = xyz.shape[1]
    traj = Trajectory(xyz, Topology([f"X{i}" for i in range(A)], ["UNK"] * A, np.zeros(A, dtype=int)))
    X, cols, per = compute_features(traj, ["Rg", "distance_pair(i=3, j=90)", "contacts_pair(i=3, j=90, rcut=0.9)"])
    x64 = xyz.astype(np.float64)
    rg = np.sqrt(((x64 - x64.mean(axis=1, keepdims=True)) ** 2).sum(axis=2).mean(axis=1))
    d = np.linalg.norm(x64[:, 90] - x64[:, 3], axis=1)
    np.testing.assert_allclose(X[:, 0], rg, rtol=2e-6)
    np.testing.assert_allclose(X[:, 1], d, rtol=2e-6)
    far = np.abs(d - 0.9) > 1e-5                       # away from the threshold the indicator is exact
    np.testing.assert_array_equal(X[far, 2], (d[far] <= 0.9).astype(float))
    assert 0 < X[:, 2].sum() < X.shape[0]
    assert cols == ["Rg", "dist:atoms:3-90", "contacts_pair"] and per.tolist() == [False, False, False]
    with pytest.raises(ValueError):
        get_feature("contacts_pair").compute(traj, i=0, j=1, rcut=0.0)
    with pytest.raises(ValueError):
        get_feature("distance_pair").compute(traj, i=0, j=A)


def test_chi1_and_backbone_torsions(engine, tmp_path):
    """chi1 = N-CA-CB-gamma for residues with a gamma atom (mdtraj.compute_chi1's patterns CG, CG1, SG, OG,
    OG1), residue order; "backbone_torsions" = [phi | psi | chi1] (S/features/featurize.py:55-62)."""
    from pmarlo_amd.features import featurize_trajectory, get_feature
    from pmarlo_amd.io import load_pdb

    rng = np.random.default_rng(5)
    residues = [("GLY", ["N", "CA", "C", "O"]), ("SER", ["N", "CA", "C", "O", "CB", "OG"]),
                ("VAL", ["N", "CA", "C", "O", "CB", "CG1", "CG2"]), ("ALA", ["N", "CA", "C", "O", "CB"]),
                ("THR", ["N", "CA", "C", "O", "CB", "OG1", "CG2"]), ("CYS", ["N", "CA", "C", "O", "CB", "SG"]),
                ("LEU", ["N", "CA", "C", "O", "CB", "CG", "CD1", "CD2"])]
    lines, names, serial = [], [], 1
    for model in range(4):
        lines.append(f"MODEL     {model + 1:4d}")
        for r, (resn, atoms) in enumerate(residues):
            for a in atoms:
                x, y, z = rng.normal(scale=4.0, size=3) + 3.0 * r
                lines.append(f"ATOM  {serial % 100000:5d} {a:<4s} {resn} A{r + 1:4d}    {x:8.3f}{y:8.3f}{z:8.3f}  1.00  0.00")
                serial += 1
                if model == 0:
                    names.append((r, a))
        lines.append("ENDMDL")
    path = tmp_path / "pep.pdb"
    path.write_text("\n".join(lines) + "\nEND\n")
    traj = load_pdb(path)
    assert traj.n_frames == 4
    at = {ra: i for i, ra in enumerate(names)}
    want_quads = [[at[(r, "N")], at[(r, "CA")], at[(r, "CB")], at[(r, g)]]
                  for r, g in ((1, "OG"), (2, "CG1"), (4, "OG1"), (5, "SG"), (6, "CG"))]
    np.testing.assert_array_equal(traj.topology.chi1_indices(), want_quads)
    fc = get_feature("chi1")
    got = fc.compute(traj)
    want = npport.dihedrals(traj.xyz, np.asarray(want_quads))
    np.testing.assert_allclose(got, want, atol=2e-5)
    assert fc.labels == ["chi1:res1", "chi1:res2", "chi1:res4", "chi1:res5", "chi1:res6"] and fc.is_periodic().all()
    bt = featurize_trajectory(traj, "backbone_torsions")
    pp = featurize_trajectory(traj, "phi_psi")
    assert bt.shape == (4, pp.shape[1] + 5)
    np.testing.assert_array_equal(bt[:, :pp.shape[1]], pp)
    np.testing.assert_allclose(bt[:, pp.shape[1]:], want, atol=2e-5)
