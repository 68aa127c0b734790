"""The adapter INTEGRATION.md section 1 describes, exercised against the reference's own registry: the engine's feature
objects are registered on ``pmarlo.features`` ("last registration wins", S/features/base.py:24-33), looked up through the
reference's ``get_feature`` and its spec parser, and a ``compute`` call dispatched that way reaches the HIP engine (which,
without a GPU, must fail loudly rather than fall back).  Runs where the reference is present (this container); the GPU
box has no /root/reference and skips."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

REF = Path("/root/reference/src")
pytestmark = pytest.mark.skipif(not (REF / "pmarlo" / "features" / "base.py").exists(), reason="reference not present")


@pytest.fixture()
def ref_features():
    sys.path.insert(0, str(REF))
    try:
        import pmarlo.features.base as ref_base
        saved = dict(ref_base.FEATURE_REGISTRY)
        yield ref_base
        ref_base.FEATURE_REGISTRY.clear()
        ref_base.FEATURE_REGISTRY.update(saved)
    finally:
        sys.path.remove(str(REF))


def _engine_objects():
    from pmarlo_amd.features import builtins as gpu
    return [gpu.PhiPsiFeature(), gpu.Chi1Feature(), gpu.DistanceFeature(), gpu.AngleFeature(), gpu.DihedralFeature(),
            gpu.RadiusOfGyrationFeature(), gpu.DistancePairFeature(), gpu.ContactsPairFeature(), gpu.SASAFeature(),
            gpu.HBondsCountFeature(), gpu.SecondaryStructureFractionFeature()]


def test_engine_features_register_on_the_reference_registry(ref_features):
    objs = _engine_objects()
    for obj in objs:
        ref_features.register_feature(obj)
    for obj in objs:
        assert ref_features.get_feature(obj.name) is obj
        assert ref_features.get_feature(obj.name.upper()) is obj            # the reference's lookup is case-insensitive
        assert callable(obj.compute) and isinstance(obj.is_periodic(), np.ndarray)   # the FeatureComputer protocol


def test_reference_spec_parser_resolves_to_engine_objects(ref_features):
    """`parse_feature_spec` of the reference names the registry keys the engine's objects carry."""
    for obj in _engine_objects():
        ref_features.register_feature(obj)
    for spec, want in [("phi_psi", "phi_psi"), ("Rg", "rg"), ("dist:AtomPair(3, 17)", "distance_pair"),
                       ("contacts:Pair(1, 2)", "contacts_pair")]:
        name, kwargs = ref_features.parse_feature_spec(spec)
        assert name.lower() == want
        assert ref_features.get_feature(name).__class__.__module__ == "pmarlo_amd.features.builtins"
    name, kwargs = ref_features.parse_feature_spec("dist:AtomPair(3, 17)")
    assert kwargs == {"i": 3, "j": 17}


def test_dispatch_through_the_reference_reaches_the_engine(ref_features):
    """No GPU here: a compute call routed through the reference's registry must end in the engine's loud failure
    (device / library missing), not in a CPU result."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_api.py")
    from pmarlo_amd import _lib

    ref_features.register_feature(_engine_objects()[5])                     # Rg
    fc = ref_features.get_feature("rg")

    class _Traj:            # the two attributes the engine's features read from an mdtraj-like trajectory
        def __init__(self):
            self.xyz = np.zeros((4, 3, 3), np.float32)
            self.n_frames, self.n_atoms = 4, 3
            self.topology = None

    with pytest.raises((_lib.MsmError, RuntimeError, OSError)):
        fc.compute(_Traj())
