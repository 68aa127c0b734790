"""DCD reader (pmarlo_amd.io.dcd): hand-assembled files per the published CHARMM/NAMD layout,
round trips, either byte order, unit cells, striding / chunking / atom slices.  CPU only."""
import struct

import numpy as np
import pytest

from pmarlo_amd.io import DCDFile, Topology, Trajectory, iterload, load_dcd, write_dcd


def _hand_dcd(path, xyz_A, order="<", cell=None, n_frames_in_header=None):
    """Bytes assembled independently of write_dcd (struct.pack, record by record)."""
    n_frames, n_atoms = xyz_A.shape[:2]
    ic = [0] * 20
    ic[0] = n_frames if n_frames_in_header is None else n_frames_in_header
    ic[1], ic[2], ic[19] = 0, 1, 24
    ic[10] = 1 if cell is not None else 0
    with open(path, "wb") as fh:
        fh.write(struct.pack(order + "i4s9if10ii", 84, b"CORD", *ic[:9], 2.0, *ic[10:20], 84))
        fh.write(struct.pack(order + "ii80si", 84, 1, b"REMARKS test".ljust(80), 84))
        fh.write(struct.pack(order + "iii", 4, n_atoms, 4))
        for f in range(n_frames):
            if cell is not None:
                fh.write(struct.pack(order + "i6di", 48, *cell[f], 48))
            for ax in range(3):
                fh.write(struct.pack(order + "i", 4 * n_atoms))
                fh.write(struct.pack(order + f"{n_atoms}f", *xyz_A[f, :, ax]))
                fh.write(struct.pack(order + "i", 4 * n_atoms))


@pytest.mark.parametrize("order", ["<", ">"])
def test_hand_assembled_file_known_values(tmp_path, order):
    xyz = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3) + 0.5
    p = tmp_path / "a.dcd"
    _hand_dcd(p, xyz, order)
    f = DCDFile(p)
    assert (f.n_frames, f.n_atoms) == (2, 3)
    assert f.layout.titles == ["REMARKS test"] and f.layout.delta == 2.0
    got, cell = f.read()
    assert cell is None and got.dtype == np.float32
    np.testing.assert_allclose(got, xyz / 10.0, rtol=3e-7)   # Angstrom -> nm (float32 multiply by 0.1)


def test_unit_cell_record_and_cosine_angles(tmp_path):
    xyz = np.random.default_rng(0).normal(size=(3, 5, 3)).astype(np.float32)
    cell_disk = np.array([[30.0, 90.0, 40.0, 80.0, 70.0, 50.0],          # a, gamma, b, beta, alpha, c
                          [30.0, 0.0, 40.0, 0.5, 0.0, 50.0],             # old CHARMM: cosines
                          [31.0, 90.0, 41.0, 90.0, 90.0, 51.0]])
    p = tmp_path / "c.dcd"
    _hand_dcd(p, xyz, "<", cell=cell_disk)
    got, cell = DCDFile(p).read()
    np.testing.assert_allclose(got, xyz / 10.0, rtol=1e-6)
    np.testing.assert_allclose(cell[0], [3.0, 4.0, 5.0, 70.0, 80.0, 90.0])
    np.testing.assert_allclose(cell[1], [3.0, 4.0, 5.0, 90.0, 60.0, 90.0], atol=1e-12)
    np.testing.assert_allclose(cell[2], [3.1, 4.1, 5.1, 90.0, 90.0, 90.0])


def test_round_trip_stride_atoms_chunks(tmp_path):
    rng = np.random.default_rng(3)
    xyz = rng.normal(size=(103, 17, 3)).astype(np.float32)
    p = tmp_path / "r.dcd"
    write_dcd(p, xyz)
    full = load_dcd(p)
    assert full.n_frames == 103 and full.n_atoms == 17
    np.testing.assert_allclose(full.xyz, xyz, rtol=2e-6, atol=1e-7)      # nm -> A (f32) -> nm
    sel = [0, 5, 16, 3]
    sub = load_dcd(p, stride=4, atom_indices=sel)
    np.testing.assert_array_equal(sub.xyz, full.xyz[::4][:, sel])
    chunks = list(iterload(p, stride=3, chunk=10))
    assert [c.n_frames for c in chunks] == [10, 10, 10, 5]
    np.testing.assert_array_equal(np.concatenate([c.xyz for c in chunks]), full.xyz[::3])
    big = tmp_path / "b.dcd"
    write_dcd(big, xyz, big_endian=True)
    np.testing.assert_array_equal(load_dcd(big).xyz, full.xyz)


def test_header_without_frame_count_and_truncation(tmp_path):
    xyz = np.ones((4, 2, 3), dtype=np.float32)
    p = tmp_path / "n.dcd"
    _hand_dcd(p, xyz, n_frames_in_header=0)            # writers that never patched NSET
    assert DCDFile(p).n_frames == 4
    data = p.read_bytes()
    q = tmp_path / "t.dcd"
    q.write_bytes(data[:-10])                          # last frame incomplete
    assert DCDFile(q).n_frames == 3


def test_topology_attachment_and_errors(tmp_path):
    xyz = np.zeros((2, 3, 3), dtype=np.float32)
    p = tmp_path / "t.dcd"
    write_dcd(p, xyz)
    top = Topology(["N", "CA", "C"], ["ALA"] * 3, np.zeros(3, dtype=int), ["A"] * 3)
    tr = load_dcd(p, top=top, atom_indices=[1])
    assert tr.topology.atom_names == ["CA"] and tr.n_atoms == 1
    with pytest.raises(ValueError):
        load_dcd(p, top=Topology(["N"], ["ALA"], np.zeros(1, dtype=int)))
    with pytest.raises(ValueError):
        load_dcd(p, atom_indices=[7])
    bad = tmp_path / "x.dcd"
    bad.write_bytes(b"not a dcd file at all, but long enough to hold the ninety-two header bytes it would need....")
    with pytest.raises(ValueError):
        DCDFile(bad)
    assert isinstance(load_dcd(p, top=Trajectory(xyz, Topology(["a", "b", "c"], ["X"] * 3, np.zeros(3, dtype=int)))),
                      Trajectory)


def test_capped_dipeptide_is_protein_throughout(tmp_path):
    """ACE-ALA-NME, the reference's standard test system: mdtraj's residue tables count the caps as protein, so
    `protein`, `backbone` and `sidechain` cover them (S/utils/mdtraj.py:31-44 hands these strings to mdtraj)."""
    from pmarlo_amd.io.pdb import load_pdb

    atoms = [("HH31", "ACE", 1), ("CH3", "ACE", 1), ("HH32", "ACE", 1), ("HH33", "ACE", 1), ("C", "ACE", 1), ("O", "ACE", 1),
             ("N", "ALA", 2), ("H", "ALA", 2), ("CA", "ALA", 2), ("HA", "ALA", 2), ("CB", "ALA", 2), ("HB1", "ALA", 2),
             ("HB2", "ALA", 2), ("HB3", "ALA", 2), ("C", "ALA", 2), ("O", "ALA", 2), ("N", "NME", 3), ("H", "NME", 3),
             ("CH3", "NME", 3), ("HH31", "NME", 3), ("HH32", "NME", 3), ("HH33", "NME", 3), ("O", "HOH", 4), ("H1", "HOH", 4)]
    lines = []
    for i, (nm, res, seq) in enumerate(atoms):
        el = "H" if nm[0] == "H" else nm[0]
        lines.append(f"ATOM  {i + 1:5d} {nm:<4s} {res:>3s} A{seq:4d}    {0.1 * i:8.3f}{0.0:8.3f}{0.0:8.3f}  1.00  0.00          {el:>2s}")
    p = tmp_path / "ala2.pdb"
    p.write_text("\n".join(lines) + "\nEND\n")
    top = load_pdb(p).topology
    assert list(top.select("protein")) == list(range(22))
    assert list(top.select("water")) == [22, 23]
    names = [a[0] for a in atoms]
    assert list(top.select("backbone")) == [i for i in range(22) if names[i] in ("N", "CA", "C", "O")]
    assert list(top.select("sidechain")) == [i for i in range(22) if names[i] not in ("N", "CA", "C", "O", "HA", "H")]
    assert list(top.select("protein and name CA")) == [8]
