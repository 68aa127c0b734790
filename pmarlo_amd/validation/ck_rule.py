"""Chapman-Kolmogorov guardrail: mirror of pmarlo.validation.ck_rule
(S/validation/ck_rule.py:14-33 config / decision, :36-47 ck_error, :50-63 sampling noise,
:71-117 decide_ck) with the matrix powers and reductions on the GPU (msm_ck_test)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Literal, Tuple

import numpy as np

from ..device import get_engine

Mode = Literal["absolute", "ess_adjusted"]

__all__ = ["CKConfig", "CKDecision", "ck_error", "decide_ck"]


@dataclass(frozen=True)
class CKConfig:
    mode: Mode = "ess_adjusted"
    absolute: float = 0.15
    min_pass_fraction: float = 0.8
    per_lag_cap: float = 0.35
    k_steps: Tuple[int, ...] = (2, 3, 4)
    sigma_mult: float = 3.0


@dataclass(frozen=True)
class CKDecision:
    pass_fraction: float
    per_lag: Dict[int, Dict[str, float]]
    passed: bool
    reason: str


def _check_square_pair(P_tau: np.ndarray, P_k_tau: np.ndarray) -> None:
    if P_tau.shape != P_k_tau.shape or P_tau.ndim != 2 or P_tau.shape[0] != P_tau.shape[1]:
        raise ValueError("P_tau and P_k_tau must be square matrices of identical shape.")


def ck_error(P_tau: np.ndarray, P_k_tau: np.ndarray, k: int) -> float:
    """RMS matrix error between P(tau)^k and P(k tau)."""
    P_tau = np.ascontiguousarray(P_tau, dtype=np.float64)
    P_k_tau = np.ascontiguousarray(P_k_tau, dtype=np.float64)
    _check_square_pair(P_tau, P_k_tau)
    eng = get_engine()
    mse, _ = eng.ck_test(eng.to_device(P_tau), eng.to_device(P_k_tau[None]), [int(k)])
    return float(np.sqrt(mse[0]))


def decide_ck(P_taus: Dict[int, np.ndarray], P_ktaus: Dict[int, np.ndarray],
              row_counts_by_lag: Dict[int, np.ndarray], cfg: CKConfig) -> CKDecision:
    """Per factor k: error = RMS(P_tau^k - P_ktau) against cfg.absolute, or against
    min(per_lag_cap, sigma_mult * multinomial noise RMS of P_ktau) in "ess_adjusted" mode."""
    if cfg.mode not in ("absolute", "ess_adjusted"):
        raise ValueError(f"Unknown CK mode: {cfg.mode}")
    eng = get_engine()
    per_lag: Dict[int, Dict[str, float]] = {}
    total = passes = 0
    for k in cfg.k_steps:
        if k not in P_taus or k not in P_ktaus or k not in row_counts_by_lag:
            continue
        total += 1
        P = np.ascontiguousarray(P_taus[k], dtype=np.float64)
        Pk = np.ascontiguousarray(P_ktaus[k], dtype=np.float64)
        _check_square_pair(P, Pk)
        rows = np.ascontiguousarray(row_counts_by_lag[k], dtype=np.float64)
        if rows.shape[0] != P.shape[0]:
            raise ValueError("counts length must equal number of states.")
        want_noise = cfg.mode == "ess_adjusted"
        mse, noise = eng.ck_test(eng.to_device(P), eng.to_device(Pk[None]), [int(k)],
                                 rowcounts=eng.to_device(rows[None]) if want_noise else None)
        err = float(np.sqrt(mse[0]))
        if want_noise:
            nz = float(noise[0])
            thr = float(min(cfg.per_lag_cap, cfg.sigma_mult * nz))
        else:
            nz, thr = float("nan"), cfg.absolute
        ok = err <= thr
        passes += int(ok)
        per_lag[k] = {"error": err, "threshold": thr, "noise_rms": nz, "pass": float(ok)}
    frac = passes / total if total > 0 else 0.0
    passed = frac >= cfg.min_pass_fraction
    reason = (f"CK guardrail {'PASSED' if passed else 'FAILED'}: {passes}/{total} lags within threshold "
              f"(pass_fraction={frac:.2f}, mode={cfg.mode}, cap={cfg.per_lag_cap}).")
    return CKDecision(pass_fraction=frac, per_lag=per_lag, passed=passed, reason=reason)
