"""expected_pairs: mirror of pmarlo.analysis.counting (S/analysis/counting.py:10-68). Host arithmetic on
a handful of segment lengths."""

from __future__ import annotations

from typing import Iterable, Sequence

__all__ = ["expected_pairs"]


def expected_pairs(lengths: Iterable[int] | Sequence[int], tau: int,
                   stride: int | Iterable[int] | Sequence[int] = 1) -> int:
    if tau < 0:
        raise ValueError("tau must be non-negative")
    length_list = [int(v) for v in lengths]
    if any(v < 0 for v in length_list):
        raise ValueError("lengths must be non-negative")
    if not length_list or not any(length_list):
        return 0
    if isinstance(stride, (str, bytes)):
        raise TypeError("stride must be an integer or iterable of integers")
    stride_values = [int(v) for v in stride] if isinstance(stride, Iterable) else [int(stride)]
    if not stride_values:
        raise ValueError("stride iterable must not be empty")
    if any(v <= 0 for v in stride_values):
        raise ValueError("stride values must be positive")
    total = 0
    for idx, length in enumerate(length_list):
        eff = length - tau
        if length <= 0 or eff <= 0:
            continue
        step = stride_values[idx] if idx < len(stride_values) else stride_values[-1]
        total += 1 + (eff - 1) // step
    return total
