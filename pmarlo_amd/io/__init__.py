"""Trajectory input for the featurizer (mdtraj is not a dependency here): PDB and DCD."""
from .dcd import DCDFile, iterload, load_dcd, write_dcd  # noqa: F401
from .pdb import Topology, Trajectory, load_pdb  # noqa: F401
