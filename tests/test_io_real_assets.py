"""Input side against the reference's own fixture files (tests/_assets/traj.dcd + 3gd8-fixed.pdb, which its tests load
through mdtraj: tests/conftest.py:208-229).  tests/golden/real_assets.npz holds the DCD header + its first four frames
byte for byte, the PDB text, and the facts an independent struct / column parse gives (tests/golden/make_golden.py);
the readers and the selection language of pmarlo_amd.io must agree with them."""

from __future__ import annotations

import logging

import numpy as np
import pytest

from pmarlo_amd.io import dcd as dcdio
from pmarlo_amd.io.pdb import load_pdb
from pmarlo_amd.utils.mdtraj import load_mdtraj_topology, resolve_atom_selection


@pytest.fixture(scope="module")
def assets(tmp_path_factory, golden):
    g = golden("real_assets.npz")
    d = tmp_path_factory.mktemp("assets")
    (d / "traj4.dcd").write_bytes(bytes(g["dcd_bytes"]))
    (d / "3gd8-fixed.pdb").write_bytes(bytes(g["pdb_text"]))
    return g, d


def _txt(a):
    return bytes(a).decode()


def test_dcd_header_and_frames_of_the_reference_asset(assets):
    g, d = assets
    f = dcdio.DCDFile(d / "traj4.dcd")
    lay = f.layout
    assert f.n_atoms == int(g["dcd_natoms"]) == 3350 and f.n_frames == 4
    assert lay.has_cell == bool(g["dcd_has_cell"]) and not lay.has_4d and lay.order == "<"
    assert lay.istart == int(g["dcd_istart"]) and lay.nsavc == int(g["dcd_nsavc"])
    assert np.float32(lay.delta) == g["dcd_delta"]
    assert lay.titles and lay.titles[0].startswith(_txt(g["dcd_title"]))
    xyz, cell = f.read()
    assert xyz.shape == (4, 3350, 3) and xyz.dtype == np.float32
    # coordinates: nm in memory, Angstrom on disk
    np.testing.assert_allclose(xyz[:, :8] * 10.0, g["dcd_xyz_angstrom_head"], rtol=2e-7)
    np.testing.assert_allclose(xyz[:, -8:] * 10.0, g["dcd_xyz_angstrom_tail"], rtol=2e-7)
    np.testing.assert_allclose(xyz.astype(np.float64).sum(axis=1) * 10.0, g["dcd_xyz_sums"], rtol=1e-6)
    np.testing.assert_allclose((xyz.astype(np.float64) ** 2).sum(axis=1) * 100.0, g["dcd_xyz_sq_sums"], rtol=1e-6)
    # unit cell record: a, gamma, b, beta, alpha, c on disk -> a, b, c (nm), alpha, beta, gamma (degrees)
    raw = g["dcd_cells_raw"]
    ang = raw[:, [4, 3, 1]]
    ang = np.where(np.all(np.abs(ang) <= 1.0, axis=1, keepdims=True), np.degrees(np.arccos(np.clip(ang, -1, 1))), ang)
    np.testing.assert_allclose(cell[:, :3], raw[:, [0, 2, 5]] / 10.0, rtol=1e-12)
    np.testing.assert_allclose(cell[:, 3:], ang, rtol=1e-12)
    assert abs(cell[0, 0] - 8.2058) < 1e-3 and abs(cell[0, 3] - 90.0) < 1e-6      # CRYST1 of the PDB: 82.058 A, 90 deg


def test_pdb_topology_of_the_reference_asset(assets):
    g, d = assets
    traj = load_pdb(d / "3gd8-fixed.pdb")
    top = traj.topology
    assert traj.n_atoms == int(g["pdb_natoms"]) == 3350 and traj.n_frames == 1
    assert top.n_residues == int(g["pdb_nres"])
    np.testing.assert_array_equal(top.select("name CA"), g["pdb_ca"])
    assert " ".join(top.atom_names[:12]) == _txt(g["pdb_first_names"])
    first_of_res = np.flatnonzero(np.r_[True, np.diff(top.res_index) != 0])[:8]
    assert " ".join(np.asarray(top.res_names)[first_of_res]) == _txt(g["pdb_first_resnames"])
    assert [int(top.res_seq[0]), int(top.res_seq[-1])] == g["pdb_resseq_first_last"].tolist()
    assert int(np.sum(np.asarray(top.elements) == "H")) == int(g["pdb_n_hydrogen"])
    import hashlib

    assert hashlib.sha256(np.asarray(top.res_index, np.int64).tobytes()).digest() == bytes(g["pdb_res_index_sha"])
    assert load_mdtraj_topology(d / "3gd8-fixed.pdb").n_atoms == 3350


def test_selection_language_on_the_reference_topology(assets):
    g, d = assets
    top = load_pdb(d / "3gd8-fixed.pdb").topology
    n = top.n_atoms
    sel = top.select
    names, resid = np.asarray(top.atom_names), np.asarray(top.res_index)
    elem = np.asarray(top.elements)
    allidx = np.arange(n)
    np.testing.assert_array_equal(sel("all"), allidx)
    np.testing.assert_array_equal(sel("protein"), allidx)                       # a protein-only structure
    assert sel("water").size == 0 and sel("none").size == 0
    np.testing.assert_array_equal(sel("backbone"), np.flatnonzero(np.isin(names, ["N", "CA", "C", "O"])))
    np.testing.assert_array_equal(sel("name CA CB"), np.flatnonzero(np.isin(names, ["CA", "CB"])))
    np.testing.assert_array_equal(sel("resid 0 to 9"), np.flatnonzero(resid <= 9))
    np.testing.assert_array_equal(sel("resid 3 7 11"), np.flatnonzero(np.isin(resid, [3, 7, 11])))
    np.testing.assert_array_equal(sel("resSeq 5 to 6"), np.flatnonzero((top.res_seq >= 5) & (top.res_seq <= 6)))
    np.testing.assert_array_equal(sel("index < 100"), allidx[:100])
    np.testing.assert_array_equal(sel("index 10 to 19 or index >= 3340"), np.r_[10:20, 3340:n])
    np.testing.assert_array_equal(sel("protein and not element H"), np.flatnonzero(elem != "H"))
    np.testing.assert_array_equal(sel("name CA and (resid < 5 or resid >= 218)"),
                                  np.flatnonzero((names == "CA") & ((resid < 5) | (resid >= 218))))
    np.testing.assert_array_equal(sel("not backbone && !(element == H)"),
                                  np.flatnonzero(~np.isin(names, ["N", "CA", "C", "O"]) & (elem != "H")))
    side = sel("sidechain")
    assert np.all(~np.isin(names[side], ["N", "CA", "C", "O", "HA", "H"])) and side.size > 0
    np.testing.assert_array_equal(np.union1d(sel("backbone"), sel("not backbone")), allidx)    # set algebra
    np.testing.assert_array_equal(sel("chainid 0"), allidx)
    np.testing.assert_array_equal(sel("resname GLN and name N"), np.flatnonzero((np.asarray(top.res_names) == "GLN") & (names == "N")))
    for bad in ("name", "frobnicate 3", "resid a", "(name CA", "name CA )", ""):
        with pytest.raises(ValueError):
            sel(bad)


def test_resolve_atom_selection_error_modes(assets, caplog):
    """S/utils/mdtraj.py:67-92: raise / warn / ignore, empty selections are failures, sequences of ints or digit strings."""
    g, d = assets
    top = load_pdb(d / "3gd8-fixed.pdb").topology
    assert resolve_atom_selection(top, None) is None
    ca = resolve_atom_selection(top, "name CA")
    assert ca == [int(i) for i in g["pdb_ca"]] and all(isinstance(i, int) for i in ca)
    assert resolve_atom_selection(top, [3, "7", np.int64(9)]) == [3, 7, 9]
    with pytest.raises(ValueError, match="produced no atoms"):
        resolve_atom_selection(top, "name XX")
    with pytest.raises(ValueError):
        resolve_atom_selection(top, "bogus words")
    with pytest.raises(ValueError, match="on_error"):
        resolve_atom_selection(top, "name CA", on_error="explode")
    log = logging.getLogger("test_sel")
    with caplog.at_level(logging.WARNING, logger="test_sel"):
        assert resolve_atom_selection(top, "name XX", logger=log, on_error="warn") is None
        assert resolve_atom_selection(top, ["x"], logger=log, on_error="warn") is None
    assert sum("atom selection failed" in r.message for r in caplog.records) == 2
    assert resolve_atom_selection(top, "name XX", on_error="ignore") is None
    assert resolve_atom_selection(top, [], on_error="ignore") is None


def test_iterload_with_selection_on_the_reference_asset(assets):
    """iterload(filename, top=, stride=, atom_indices=, chunk=) as S/io/trajectory.py:136-177 streams it."""
    g, d = assets
    top = load_pdb(d / "3gd8-fixed.pdb")
    ca = resolve_atom_selection(top.topology, "name CA")
    chunks = list(dcdio.iterload(d / "traj4.dcd", top=top, stride=2, atom_indices=ca, chunk=1))
    assert [c.n_frames for c in chunks] == [1, 1] and chunks[0].n_atoms == len(ca)
    full, _ = dcdio.DCDFile(d / "traj4.dcd").read()
    np.testing.assert_array_equal(chunks[1].xyz[0], full[2][ca])
    assert chunks[0].topology.atom_names == ["CA"] * len(ca)
