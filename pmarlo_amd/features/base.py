"""Feature registry: mirror of pmarlo.features.base (S/features/base.py:11-47, 129-).

``register_feature(obj)`` stores ``obj`` under ``obj.name.lower()`` (last registration
wins); ``obj.compute(traj, **kw) -> (n_frames, n_cols)``; ``obj.is_periodic() -> bool
(n_cols,)`` valid after compute; optional ``obj.labels``."""

from __future__ import annotations

from typing import Any, Dict, Protocol, Tuple, runtime_checkable

import numpy as np

__all__ = ["FeatureComputer", "FEATURE_REGISTRY", "register_feature", "get_feature", "parse_feature_spec"]


@runtime_checkable
class FeatureComputer(Protocol):
    name: str

    def compute(self, traj, **kwargs) -> np.ndarray: ...

    def is_periodic(self) -> np.ndarray: ...


FEATURE_REGISTRY: Dict[str, FeatureComputer] = {}


def register_feature(fc: FeatureComputer) -> None:
    FEATURE_REGISTRY[fc.name.lower()] = fc


def get_feature(name: str) -> FeatureComputer:
    key = name.lower()
    if key not in FEATURE_REGISTRY:
        raise KeyError(f"Unknown feature: {name}")
    return FEATURE_REGISTRY[key]


def parse_feature_spec(spec: str) -> Tuple[str, Dict[str, Any]]:
    """``"distance([0, 5])"`` -> ("distance", {"indices": [0, 5]}); ``"phi_psi"`` -> ("phi_psi", {})."""
    s = spec.strip()
    if "(" not in s:
        if ":" in s:   # namespaced forms of the structure features (S/features/base.py:160-165)
            prefix = s.split(":", 1)[0].strip().lower()
            alias = {"sasa": "sasa", "hbonds": "hbonds_count", "hbond": "hbonds_count",
                     "hbondscount": "hbonds_count", "ssfrac": "ssfrac", "secondary": "ssfrac"}.get(prefix)
            if alias:
                return alias, {}
        return s, {}
    name, rest = s.split("(", 1)
    body = rest.rsplit(")", 1)[0].strip()
    kwargs: Dict[str, Any] = {}
    if body:
        import ast

        try:
            val = ast.literal_eval(body)
            kwargs["indices"] = list(val) if isinstance(val, (list, tuple)) else val
        except (ValueError, SyntaxError):
            for part in body.split(","):
                if "=" in part:
                    k, v = part.split("=", 1)
                    try:
                        kwargs[k.strip()] = ast.literal_eval(v.strip())
                    except (ValueError, SyntaxError):
                        kwargs[k.strip()] = v.strip()
    return name.strip(), kwargs
