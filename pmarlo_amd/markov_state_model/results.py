"""Result containers (S/markov_state_model/results.py:136-147, S/utils/msm_utils.py:108-126)."""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

__all__ = ["ITSResult", "ConnectedCountResult", "MSMEstimate"]


@dataclass
class ITSResult:
    lag_times: np.ndarray = field(default_factory=lambda: np.zeros((0,), dtype=int))
    eigenvalues: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    eigenvalues_ci: np.ndarray = field(default_factory=lambda: np.zeros((0, 0, 2)))
    timescales: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    timescales_ci: np.ndarray = field(default_factory=lambda: np.zeros((0, 0, 2)))
    rates: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    rates_ci: np.ndarray = field(default_factory=lambda: np.zeros((0, 0, 2)))
    recommended_lag_window: tuple[float, float] | None = None


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
