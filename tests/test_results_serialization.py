"""Result records keep the reference's serialisation surface (S/markov_state_model/results.py:20-100,
free_energy.py:159-251; the reference's own test: tests/unit/results/test_results_serialization.py)."""

from __future__ import annotations

import json
import pickle

import numpy as np
import pytest

from pmarlo_amd.markov_state_model.free_energy import FESResult
from pmarlo_amd.markov_state_model.results import CKITSSelectionResult, ClusteringResult, ITSResult, MSMResult


def test_fes_result_roundtrip_like_the_reference_test(tmp_path):
    fes = FESResult(free_energy=np.zeros((2, 2)), xedges=np.array([0.0, 1.0, 2.0]), yedges=np.array([0.0, 1.0, 2.0]),
                    cv1_name="x", cv2_name="y", temperature=300.0)
    results = {"fes": fes}
    pkl, js = tmp_path / "analysis_results.pkl", tmp_path / "analysis_results.json"
    with pkl.open("wb") as fh:
        pickle.dump(results, fh)
    with js.open("w") as fh:
        json.dump({k: v.to_dict(metadata_only=True) for k, v in results.items()}, fh)
    loaded = pickle.load(pkl.open("rb"))
    assert isinstance(loaded["fes"], FESResult) and loaded["fes"].output_shape == (2, 2)
    assert loaded["fes"].temperature == pytest.approx(300.0)
    meta = json.load(js.open())
    assert meta["fes"]["free_energy"]["shape"] == [2, 2]
    bad = fes.to_dict()
    bad["version"] = "0"
    with pytest.raises(ValueError):
        FESResult.from_dict(bad)
    back = FESResult.from_dict(json.loads(json.dumps(fes.to_dict())))
    np.testing.assert_array_equal(back.F, fes.F)
    assert back.cv1_name == "x" and back.cv2_name == "y" and back.temperature == 300.0
    stub = FESResult.from_dict(fes.to_dict(metadata_only=True))
    assert stub.output_shape == (2, 2)
    with pytest.warns(DeprecationWarning):
        assert fes["xedges"] is fes.xedges
    with pytest.raises(TypeError):
        FESResult(xedges=[0, 1], yedges=[0, 1])


def test_base_result_dict_json_pickle(tmp_path):
    msm = MSMResult(transition_matrix=np.eye(3), count_matrix=np.arange(9.0).reshape(3, 3))
    d = msm.to_dict()
    assert d["version"] == "1.0" and d["transition_matrix"] == np.eye(3).tolist() and d["free_energies"] is None
    assert msm.to_dict(metadata_only=True)["count_matrix"] == {"shape": [3, 3], "dtype": "float64"}
    back = MSMResult.from_json(msm.to_json())
    np.testing.assert_array_equal(back.transition_matrix, msm.transition_matrix)
    assert back.output_shape == (3,)
    with pytest.raises(ValueError, match="Version mismatch"):
        MSMResult.from_dict({**d, "version": "0.9"})
    with pytest.raises(ValueError, match="Version mismatch"):
        MSMResult.from_dict({k: v for k, v in d.items() if k != "version"})
    path = tmp_path / "msm.pkl"
    msm.to_pickle(path)
    np.testing.assert_array_equal(MSMResult.from_pickle(path).count_matrix, msm.count_matrix)
    with pytest.raises(TypeError, match="Expected ClusteringResult"):
        ClusteringResult.from_pickle(path)


def test_its_and_selection_records():
    its = ITSResult(lag_times=np.array([1, 2]), eigenvalues=np.ones((2, 3)), eigenvalues_ci=np.ones((2, 3, 2)),
                    timescales=np.ones((2, 3)), timescales_ci=np.ones((2, 3, 2)), rates=np.ones((2, 3)),
                    rates_ci=np.ones((2, 3, 2)), recommended_lag_window=(1.0, 2.0))
    back = ITSResult.from_json(its.to_json())
    assert back.timescales.shape == (2, 3) and tuple(back.recommended_lag_window) == (1.0, 2.0)
    assert ITSResult().lag_times.size == 0
    sel = CKITSSelectionResult(selected_lag=5, ck_errors={5: 0.1}, its_timescales=np.ones((1, 2)), its_lag_times=np.array([5]),
                               coverage_fractions={5: 1.0}, median_counts={5: 10}, macrostate_counts={5: 3},
                               passed_sanity={5: True})
    d = sel.to_dict()
    assert d["selected_lag"] == 5 and d["diagnostics"] == {} and d["its_lag_times"] == [5]
