"""pytest configuration: `gpu` marker, repo root on sys.path, shared fixtures."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def _load(name: str):
        return np.load(GOLDEN / name, allow_pickle=False)

    return _load


@pytest.fixture(scope="session")
def engine():
    """One Engine on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    from pmarlo_amd.device import Engine

    eng = Engine(0)
    yield eng
    eng.close()
