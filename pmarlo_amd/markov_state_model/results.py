"""Result containers of the MSM path with the reference's serialisation surface.

Where the reference has ``pmarlo.markov_state_model.results`` (S/markov_state_model/results.py:20-165: ``BaseResult``
with ``to_dict / from_dict / to_json / from_json / to_pickle / from_pickle`` and a class-level ``version``, and the
result records built on it) and ``ConnectedCountResult`` (S/utils/msm_utils.py:108-126).  Same class names, field
names, defaults and wire format (``to_dict`` = the dataclass fields with arrays as nested lists, or as
``{"shape", "dtype"}`` stubs when ``metadata_only``, plus ``"version"``), so files written by either side load on
the other.  Written against that behaviour, not against the reference's text."""

from __future__ import annotations

import dataclasses
import json
import logging
import pickle
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, ClassVar, Dict, Optional

import numpy as np

__all__ = ["BaseResult", "ClusteringResult", "MSMResult", "CKResult", "ITSResult", "CKITSSelectionResult",
           "ConnectedCountResult", "MSMEstimate"]

logger = logging.getLogger("pmarlo")


def _plain(value: Any, metadata_only: bool) -> Any:
    """JSON-ready form of one field value (arrays -> lists or shape / dtype stubs, lists element by element)."""
    if isinstance(value, np.ndarray):
        return {"shape": list(value.shape), "dtype": str(value.dtype)} if metadata_only else value.tolist()
    if isinstance(value, list):
        return [_plain(v, metadata_only) for v in value]
    return value


@dataclass
class BaseResult:
    """Serialisation helpers shared by the result records."""

    version: ClassVar[str] = "1.0"

    # ---- dict / JSON -------------------------------------------------------------------------------------------
    def to_dict(self, metadata_only: bool = False) -> Dict[str, Any]:
        out = {name: _plain(value, metadata_only) for name, value in dataclasses.asdict(self).items()}
        out["version"] = self.version
        return out

    @classmethod
    def _check_version(cls, found: Any) -> None:
        if found != cls.version:
            logger.error("Version mismatch when loading %s: %s != %s", cls.__name__, found, cls.version)
            raise ValueError(f"Version mismatch: {found} != {cls.version}")

    @classmethod
    def from_dict(cls, data: Dict[str, Any]):
        data = dict(data)
        cls._check_version(data.pop("version", None))
        array_fields = {f.name for f in dataclasses.fields(cls) if f.type in (np.ndarray, "np.ndarray")}
        kwargs = {k: (np.asarray(v) if k in array_fields and isinstance(v, list) else v) for k, v in data.items()}
        return cls(**kwargs)

    def to_json(self, metadata_only: bool = False) -> str:
        return json.dumps(self.to_dict(metadata_only=metadata_only))

    @classmethod
    def from_json(cls, text: str):
        return cls.from_dict(json.loads(text))

    # ---- pickle ------------------------------------------------------------------------------------------------
    def to_pickle(self, path: Path) -> None:
        with Path(path).open("wb") as fh:
            pickle.dump(self, fh)

    @classmethod
    def from_pickle(cls, path: Path):
        with Path(path).open("rb") as fh:
            obj = pickle.load(fh)
        if not isinstance(obj, cls):
            raise TypeError(f"Expected {cls.__name__}, got {type(obj).__name__}")
        cls._check_version(getattr(obj, "version", None))
        return obj


@dataclass
class ClusteringResult(BaseResult):
    """Assignments and centres of a clustering run (the record of ``results.py``; ``cluster_microstates`` returns
    ``pmarlo_amd.markov_state_model.clustering.ClusteringResult`` as the reference's does)."""

    assignments: np.ndarray
    centers: np.ndarray


@dataclass
class MSMResult(BaseResult):
    transition_matrix: np.ndarray
    count_matrix: np.ndarray
    free_energies: Optional[np.ndarray] = None
    stationary_distribution: Optional[np.ndarray] = None

    @property
    def output_shape(self) -> tuple[int, ...]:
        return (self.transition_matrix.shape[0],)


@dataclass
class CKResult(BaseResult):
    lag_times: np.ndarray
    timescales: np.ndarray


@dataclass
class ITSResult(BaseResult):
    """Implied timescales with confidence intervals (``compute_implied_timescales``)."""

    lag_times: np.ndarray = field(default_factory=lambda: np.zeros((0,), dtype=int))
    eigenvalues: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    eigenvalues_ci: np.ndarray = field(default_factory=lambda: np.zeros((0, 0, 2)))
    timescales: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    timescales_ci: np.ndarray = field(default_factory=lambda: np.zeros((0, 0, 2)))
    rates: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    rates_ci: np.ndarray = field(default_factory=lambda: np.zeros((0, 0, 2)))
    recommended_lag_window: Optional[tuple[float, float]] = None


@dataclass
class CKITSSelectionResult(BaseResult):
    """Outcome of the CK + ITS lag selection (``select_optimal_lag_ck_its``)."""

    selected_lag: int
    ck_errors: Dict[int, float]
    its_timescales: np.ndarray
    its_lag_times: np.ndarray
    coverage_fractions: Dict[int, float]
    median_counts: Dict[int, int]
    macrostate_counts: Dict[int, int]
    passed_sanity: Dict[int, bool]
    diagnostics: Dict[str, Any] = field(default_factory=dict)


@dataclass
class ConnectedCountResult:
    counts: np.ndarray
    active: np.ndarray

    def to_dict(self):
        return {"counts": self.counts.tolist(), "active": self.active.tolist()}


@dataclass
class MSMEstimate:
    """What EstimationMixin._finalize_transition_and_stationary sets on the model."""

    count_matrix: np.ndarray
    transition_matrix: np.ndarray
    stationary_distribution: np.ndarray
    active: np.ndarray
    free_energies: np.ndarray | None = None
