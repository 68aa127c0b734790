"""Host-side lag helpers of the ITS scan: plateau window, lag choice, lag ladder.

detect_timescale_plateau / tail bounds / default lags are pinned by tests/golden/its_helpers.npz (made
by importing the reference's ITSMixin).  select_lag_from_its and candidate_lag_ladder live in reference
modules that need deeptime to import: their expectations below are worked by hand from the reference's
documented rules (S/markov_state_model/_msm_utils.py:302-399, S/utils/msm_utils.py:21-105)."""
import numpy as np
import pytest

from pmarlo_amd.markov_state_model import (
    DEFAULT_ITS_LAGS,
    candidate_lag_ladder,
    detect_timescale_plateau,
    select_lag_from_its,
)


def test_plateau_windows_match_reference(golden):
    g = golden("its_helpers.npz")
    np.testing.assert_array_equal(np.asarray(DEFAULT_ITS_LAGS), g["default_lags"])
    lags, want = g["plateau_lags"], g["plateau_windows"]
    row = 0
    for ts in g["plateau_series"]:
        for m, eps in ((2, 0.05), (3, 0.1), (4, 0.2), (1, 0.0)):
            got = detect_timescale_plateau(lags, ts, m, eps)
            if np.isnan(want[row, 0]):
                assert got is None
            else:
                assert got == (want[row, 0], want[row, 1])
            row += 1
    assert detect_timescale_plateau(np.array([]), np.zeros((0, 2)), 2, 0.1) is None


def test_select_lag_from_its_rules():
    lags = np.array([1, 2, 5, 10, 20, 30, 50, 100])
    rising = np.array([2.0, 5.0, 11.0, 20.0, 31.0, 34.0, 35.0, 35.5])
    # first change < 15 % from index 3 on: 31 -> 34 (9.7 %), confirmed by 34 -> 35 (2.9 % < 22.5 %)
    assert select_lag_from_its(lags, rising[:, None]) == 30
    # the candidate at index 5 is not confirmed (next step +60 %), index 6 is not below 15 %, the last
    # point has no successor and passes on its own change (4 %)
    fluke = np.array([2.0, 5.0, 11.0, 20.0, 31.0, 34.0, 54.4, 56.5])
    assert select_lag_from_its(lags, fluke) == 100
    # nothing settles: the largest timescale of the second half
    wild = np.array([1.0, 2.0, 4.0, 8.0, 16.0, 40.0, 20.0, 90.0])
    assert select_lag_from_its(lags, wild) == 100
    wild[-1] = np.nan
    assert select_lag_from_its(lags, wild) == 30
    assert select_lag_from_its(np.array([]), np.array([])) == 10
    assert select_lag_from_its(lags, np.full(8, np.nan)) == 10
    # min_lag_idx beyond the scan falls back to a quarter of its length
    assert select_lag_from_its(lags[:3], np.array([10.0, 10.5, 10.6]), min_lag_idx=7) == 2


def test_candidate_lag_ladder():
    assert candidate_lag_ladder() == [1, 2, 3, 5, 8, 10, 15, 20, 30, 40, 50, 75, 80, 100, 150, 160, 200]
    assert candidate_lag_ladder(10, 100) == [10, 15, 20, 30, 40, 50, 75, 80, 100]
    assert candidate_lag_ladder(1, 200, 1) == [1] and candidate_lag_ladder(1, 200, 2) == [1, 200]
    # 17 values, 5 picks: round(q * 4.0) = 0, 4, 8, 12, 16
    assert candidate_lag_ladder(1, 200, 5) == [1, 8, 30, 80, 200]
    assert candidate_lag_ladder(1, 2000, 100)[-1] == 2000
    for bad in (dict(min_lag=0), dict(min_lag=5, max_lag=4), dict(n_candidates=0), dict(min_lag=201, max_lag=299)):
        with pytest.raises(ValueError):
            candidate_lag_ladder(**bad)


def test_oracle_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10 (the generator behind the posterior samples)."""
    from oracle import npport

    kat = [(0, [0, 0, 0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
           ((0xFFFFFFFF << 32) | 0xFFFFFFFF, [0xFFFFFFFF] * 4, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
           ((0x299F31D0 << 32) | 0xA4093822, [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344],
            [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1])]
    for key, ctr, want in kat:
        np.testing.assert_array_equal(npport.philox4x32(key, np.asarray(ctr, np.uint32)), np.asarray(want, np.uint32))


def test_format_lag_window_ps():
    from pmarlo_amd.markov_state_model.its import format_lag_window_ps

    assert format_lag_window_ps((2.0, 10.5)) == "2.000–10.500 ps"
