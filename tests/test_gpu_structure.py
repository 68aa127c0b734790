"""Structure features on the device (csrc/structure.hip through the C ABI) against the CPU restatements of
oracle/npport.py: Shrake-Rupley areas (bit for bit: the same fp32 operations), Baker-Hubbard presence counts and the
Kabsch-Sander codes, on constructed geometries and on the reference's own asset (3gd8-fixed.pdb + the first frames of
traj.dcd, tests/golden/real_assets.npz).  Call sites served: S/features/builtins.py:171-250."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import npport
from pmarlo_amd.features import structure as st
from pmarlo_amd.features.base import get_feature
from pmarlo_amd.io import dcd as dcdio
from pmarlo_amd.io.pdb import Trajectory, load_pdb

from .test_structure_oracle import _ideal_helix, _water_free_dipeptide

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def protein(tmp_path_factory, golden):
    g = golden("real_assets.npz")
    d = tmp_path_factory.mktemp("assets")
    (d / "traj4.dcd").write_bytes(bytes(g["dcd_bytes"]))
    (d / "p.pdb").write_bytes(bytes(g["pdb_text"]))
    pdb = load_pdb(d / "p.pdb")
    xyz, _ = dcdio.DCDFile(d / "traj4.dcd").read()
    return Trajectory(np.concatenate([pdb.xyz, xyz], axis=0), pdb.topology)      # 5 frames, 3350 atoms


def test_sasa_matches_the_oracle_bit_for_bit(engine, protein):
    rng = np.random.default_rng(5)
    A = 300
    xyz = (rng.random((3, A, 3)) * 1.6).astype(np.float32)                         # dense: ~60 neighbours per atom
    radii = rng.choice(np.float32([0.26, 0.31, 0.295, 0.292, 0.32]), size=A)
    got = engine.featurize_sasa(engine.to_device(xyz), radii, st.sphere_points(960)).to_host()
    want = npport.shrake_rupley_atoms(xyz, radii, 960)
    np.testing.assert_array_equal(got, want)
    # fewer points, a single atom, two atoms
    got = engine.featurize_sasa(engine.to_device(xyz[:1, :2]), radii[:2], st.sphere_points(100)).to_host()
    np.testing.assert_array_equal(got, npport.shrake_rupley_atoms(xyz[:1, :2], radii[:2], 100))


def test_sasa_of_the_reference_protein(engine, protein):
    sub = Trajectory(protein.xyz[:2], protein.topology)
    per_res = st.shrake_rupley(sub, mode="residue")
    assert per_res.shape == (2, protein.topology.n_residues) and per_res.dtype == np.float32
    el = [e.capitalize() for e in protein.topology.elements]
    radii = np.float32([st.ATOMIC_RADII[e] for e in el]) + np.float32(0.14)
    # oracle on the first 40 residues' atoms of frame 0 would change the neighbourhoods: check whole-protein atoms of
    # a slab instead: the oracle gets ALL atoms, compared on every 25th
    want = npport.shrake_rupley_atoms(sub.xyz[:1], radii, 960)
    got = st.shrake_rupley(Trajectory(sub.xyz[:1], protein.topology), mode="atom")
    np.testing.assert_array_equal(got, want)
    total = get_feature("sasa").compute(sub)
    assert total.shape == (2, 1) and total.dtype == np.float64
    np.testing.assert_allclose(total[:, 0], per_res.sum(axis=1), rtol=1e-6)
    assert 80.0 < total[0, 0] < 200.0          # nm^2: a 223-residue membrane protein monomer


def test_hbond_presence_and_count(engine, protein):
    traj = _water_free_dipeptide()
    trip = st.hbond_triplets(traj)
    xyz = traj.xyz.copy()
    xyz[:2, 3] = [0.12, 0.17, 0.0]
    got = engine.hbond_presence(engine.to_device(xyz), trip, 0.25, np.float32(2 * np.pi / 3))
    np.testing.assert_array_equal(got, npport.baker_hubbard_presence(xyz, trip))
    np.testing.assert_array_equal(got, [3, 0])
    # the real protein: thousands of triplets, every frame
    trip = st.hbond_triplets(protein)
    assert len(trip) > 10000
    got = engine.hbond_presence(engine.to_device(protein.xyz), trip, 0.25, np.float32(2 * np.pi / 3))
    want = npport.baker_hubbard_presence(protein.xyz, trip)
    np.testing.assert_array_equal(got, want)
    hb = st.baker_hubbard(protein)
    np.testing.assert_array_equal(hb, trip[want / protein.n_frames > 0.1])
    assert 100 < len(hb) < 400                 # a helical 223-residue protein: roughly one backbone bond per residue
    col = get_feature("hbonds_count").compute(protein)
    assert col.shape == (protein.n_frames, 1) and (col == float(len(hb))).all()


def test_dssp_matches_the_oracle(engine, protein):
    n_res = 16
    xyz = np.concatenate([_ideal_helix(n_res), _ideal_helix(n_res) * np.float32(1.02)], axis=0)
    bb = np.arange(4 * n_res).reshape(n_res, 4)
    z = np.zeros(n_res, int)
    got = engine.dssp(engine.to_device(xyz), bb, z, z.astype(bool))
    np.testing.assert_array_equal(got, npport.dssp_codes(xyz, bb, z, z.astype(bool)))
    assert (got[0, 1:-1] == 1).sum() >= n_res - 4
    # the reference's protein, all five frames, full codes
    keep, table, chain, proline = st.backbone_table(protein.topology)
    got = engine.dssp(engine.to_device(protein.xyz), table, chain, proline)
    want = npport.dssp_codes(protein.xyz, table, chain, proline)
    np.testing.assert_array_equal(got, want)
    simple = st.compute_dssp(protein)
    assert simple.shape == (protein.n_frames, protein.topology.n_residues)
    assert set(np.unique(simple)) <= {"H", "E", "C", "NA"}
    full = st.compute_dssp(protein, simplified=False)
    assert (full[simple == "H"] != " ").all()
    frac = get_feature("ssfrac").compute(protein)
    assert frac.shape == (protein.n_frames, 3)
    np.testing.assert_allclose(frac.sum(axis=1), 1.0, atol=1e-12)
    assert (frac[:, 0] > 0.5).all() and (frac[:, 1] < 0.1).all()


def test_structure_features_edge_cases(engine, protein):
    """No hydrogens -> no donors -> count 0; a residue without a backbone atom or a non-protein residue -> 'NA' (counted
    in the ssfrac denominator, as the reference does); two chains: no helix across the break; zero frames."""
    from pmarlo_amd.io.pdb import Topology

    n_res = 14
    xyz = _ideal_helix(n_res)                                   # N, CA, C, O per residue
    names = ["N", "CA", "C", "O"] * n_res
    resn = [nm for r in range(n_res) for nm in [("HOH" if r == 5 else "ALA")] * 4]
    res_index = np.repeat(np.arange(n_res), 4)
    chains = ["A"] * (4 * 7) + ["B"] * (4 * 7)
    top = Topology(names, resn, res_index, chains, elements=["N", "C", "C", "O"] * n_res)
    traj = Trajectory(np.repeat(xyz, 3, axis=0), top)
    assert len(st.hbond_triplets(traj)) == 0
    np.testing.assert_array_equal(get_feature("hbonds_count").compute(traj), np.zeros((3, 1)))
    codes = st.compute_dssp(traj)
    assert codes.shape == (3, n_res) and (codes[:, 5] == "NA").all()
    keep, table, chain, proline = st.backbone_table(top)
    assert 5 not in keep and len(keep) == n_res - 1
    full = engine.dssp(engine.to_device(traj.xyz), table, chain, proline)
    np.testing.assert_array_equal(full, npport.dssp_codes(traj.xyz, table, chain, proline))
    # residue 5 is missing and the chain changes after residue 6: the two stretches are too short for a full turn pair
    frac = get_feature("ssfrac").compute(traj)
    np.testing.assert_allclose(frac.sum(axis=1), 1.0, atol=1e-12)
    # zero frames
    empty = Trajectory(np.zeros((0, 4 * n_res, 3), np.float32), top)
    assert st.shrake_rupley(empty).shape == (0, 4 * n_res)
    assert st.compute_dssp(empty).shape == (0, n_res)
    assert get_feature("sasa").compute(empty).shape == (0, 1)
    # an element without a radius: the feature returns zeros, as the reference does when mdtraj raises
    odd = Topology(names, ["ALA"] * (4 * n_res), res_index, ["A"] * (4 * n_res), elements=["Xx"] * (4 * n_res))
    np.testing.assert_array_equal(get_feature("sasa").compute(Trajectory(traj.xyz, odd)), np.zeros((3, 1)))


def test_baker_hubbard_frequency_threshold(engine):
    traj = _water_free_dipeptide()                               # 5 frames, the bond present in all
    assert len(st.baker_hubbard(traj, freq=0.1)) == 1
    xyz = traj.xyz.copy()
    xyz[:4, 3] = [0.12, 0.17, 0.0]                               # present in 1 of 5 frames only
    t2 = Trajectory(xyz, traj.topology)
    assert len(st.baker_hubbard(t2, freq=0.1)) == 1              # 0.2 > 0.1
    assert len(st.baker_hubbard(t2, freq=0.2)) == 0              # strictly greater
    assert len(st.baker_hubbard(t2, freq=0.0)) == 1
