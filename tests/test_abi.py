"""The C-ABI library builds, loads and exports every symbol include/msmhip.h declares.
No compute calls here (CPU suite)."""

from __future__ import annotations

import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared_functions() -> list[str]:
    text = (ROOT / "include" / "msmhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_functions():
    names = _declared_functions()
    assert "msm_count_transitions" in names and "msm_kmeans_assign" in names
    assert len(names) >= 20


def test_library_exports_every_declared_symbol():
    from pmarlo_amd import _lib
    from pmarlo_amd.csrc import build

    build.build()
    handle = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [n for n in _declared_functions() if not hasattr(handle, n)]
    assert not missing, f"declared in msmhip.h but not exported: {missing}"


def test_python_binding_covers_the_header():
    from pmarlo_amd import _lib

    declared = set(_declared_functions())
    bound = set(_lib.DECLARED_SYMBOLS)
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    lib = _lib.load()
    assert lib.msm_version().startswith(b"msmhip")


def test_missing_gpu_fails_loudly():
    """Without a HIP device the engine must raise, never fall back."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pmarlo_amd._lib import MsmError
    from pmarlo_amd.device import Engine

    with pytest.raises(MsmError):
        Engine(0)


def test_product_never_imports_the_oracle():
    for py in (ROOT / "pmarlo_amd").rglob("*.py"):
        src = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), py
