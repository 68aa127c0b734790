"""Minimal trajectory input for the featurizer (mdtraj is not a dependency here)."""
from .pdb import Topology, Trajectory, load_pdb  # noqa: F401
