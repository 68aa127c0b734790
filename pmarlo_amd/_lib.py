"""ctypes binding of libmsmhip.so (include/msmhip.h).

The library is loaded from ``pmarlo_amd/csrc/libmsmhip.so`` only; there is no
CPU fallback.  If the shared object is missing, or a GPU call fails, the error
is raised -- a silent fallback would void every parity claim.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

__all__ = ["lib", "load", "LIB_PATH", "check", "MsmError", "DECLARED_SYMBOLS"]

LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libmsmhip.so"

MSM_OK, MSM_ERR_INVALID, MSM_ERR_HIP, MSM_ERR_NOMEM, MSM_ERR_UNSUPPORTED, MSM_ERR_NOCONV = range(6)
MSM_F32, MSM_F64 = 0, 1


class MsmError(RuntimeError):
    """A HIP-side failure reported through the C ABI."""


_vp = C.c_void_p
_i32 = C.c_int
_i64 = C.c_int64
_sz = C.c_size_t
_f64 = C.c_double
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/msmhip.h one to one.
_PROTOTYPES: dict[str, tuple] = {
    "msm_version": (C.c_char_p, []),
    "msm_ctx_create": (_i32, [_i32, _vp, _pp]),
    "msm_ctx_destroy": (None, [_vp]),
    "msm_last_error": (C.c_char_p, [_vp]),
    "msm_device_info": (_i32, [_vp, C.c_char_p, _sz, C.POINTER(_i32), C.POINTER(_sz)]),
    "msm_malloc": (_i32, [_vp, _sz, _pp]),
    "msm_free": (_i32, [_vp, _vp]),
    "msm_memcpy_h2d": (_i32, [_vp, _vp, _vp, _sz]),
    "msm_memcpy_d2h": (_i32, [_vp, _vp, _vp, _sz]),
    "msm_memcpy_d2d": (_i32, [_vp, _vp, _vp, _sz]),
    "msm_memset": (_i32, [_vp, _vp, _i32, _sz]),
    "msm_sync": (_i32, [_vp]),
    "msm_event_create": (_i32, [_vp, _pp]),
    "msm_event_destroy": (None, [_vp]),
    "msm_event_record": (_i32, [_vp, _vp]),
    "msm_event_elapsed_ms": (_i32, [_vp, _vp, C.POINTER(C.c_float)]),
    "msm_graph_begin": (_i32, [_vp]),
    "msm_graph_end": (_i32, [_vp, _pp]),
    "msm_graph_launch": (_i32, [_vp, _vp]),
    "msm_graph_destroy": (None, [_vp]),
    "msm_featurize_distances": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _i64, _i32]),
    "msm_featurize_contacts": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, C.c_float, _vp, _i64, _i32]),
    "msm_featurize_rg": (_i32, [_vp, _vp, _i64, _i32, _vp, _i64, _i32]),
    "msm_featurize_angles": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _i64, _i32]),
    "msm_featurize_dihedrals": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _i32, _vp, _i64, _i32]),
    "msm_featurize_sasa": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp]),
    "msm_hbond_presence": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, C.c_float, C.c_float, _vp]),
    "msm_dssp": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp]),
    "msm_count_transitions": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "msm_count_transitions_weighted": (
        _i32, [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "msm_count_transitions_lagscan": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "msm_state_counts": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "msm_column_moments_partial": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp]),
    "msm_moments_finalize": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "msm_column_moments": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _i32, _vp, _vp, _vp]),
    "msm_column_minmax": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp]),
    "msm_standardise_params": (_i32, [_vp, _vp, _vp, _i32, _f64, _i32, _vp, _vp, _vp]),
    "msm_lagged_moments": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "msm_tica_solve": (_i32, [_vp, _vp, _vp, _i32, _f64, _i32, _vp, _vp, _vp, _vp]),
    "msm_lagged_moments_reversible": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "msm_lagged_moments_onesided": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "msm_onesided_tica_eigenvalues": (_i32, [_vp, _vp, _i32, _f64, _vp]),
    "msm_moments_from_lagged": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "msm_project": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _i64, _vp]),
    "msm_project_finite": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _i64, _vp]),
    "msm_eigh": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "msm_kmeans_assign": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp]),
    "msm_kmeans_fit": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _i32, C.c_uint64, _i32, _i32, _f64, _vp, _vp]),
    "msm_kmeans_fit_begin": (
        _i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _i32, C.c_uint64, _i32, _f64, _f64, _vp, _vp, _i32]),
    "msm_kmeans_accumulate": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "msm_kmeans_update": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32]),
    "msm_comm_unique_id": (_i32, [_vp, _sz]),
    "msm_comm_init": (_i32, [_vp, _i32, _i32, _vp, _sz, _pp]),
    "msm_comm_destroy": (None, [_vp]),
    "msm_comm_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_uint64)]),
    "msm_allreduce_i64": (_i32, [_vp, _vp, _sz]),
    "msm_allreduce_i64_from": (_i32, [_vp, _vp, _vp, _sz]),
    "msm_allreduce_f64": (_i32, [_vp, _vp, _sz]),
    "msm_allreduce_min_f64": (_i32, [_vp, _vp, _sz]),
    "msm_allreduce_max_f64": (_i32, [_vp, _vp, _sz]),
    "msm_broadcast": (_i32, [_vp, _vp, _sz, _i32]),
    "msm_rcp_f64": (_i32, [_vp, _vp, _sz, _vp]),
    "msm_kmeans_image_bytes": (_i32, [_i64, _i32, C.POINTER(_sz)]),
    "msm_kmeans_pack": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp]),
    "msm_kmeans_assign_packed": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "msm_kmeans_accumulate_packed": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "msm_kmeans_lloyd_pass": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "msm_kmeans_accumulate_delta": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _vp]),
    "msm_kmeans_filter_scanned": (_i32, [_vp, C.POINTER(C.c_uint64), _i32]),
    "msm_kmeans_init_plusplus": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _i32, C.c_uint64, _f64, _vp, _vp, _vp]),
    "msm_mfma_bf16_probe": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32]),
    "msm_run_lengths": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    "msm_sum_f64": (_i32, [_vp, _vp, _i64, _vp]),
    "msm_transition_matrix": (_i32, [_vp, _vp, _i32, _i32, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "msm_embed_full": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "msm_spectrum_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "msm_spectrum": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _i32, C.c_uint64, _i32, _vp, _vp, _vp,
                            _i64, _vp, _vp, _i32, _vp, _vp, _vp, _f64, _vp, _i32]),
    "msm_matrix_power": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _vp, _vp]),
    "msm_spectrum_powered": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _i32, C.c_uint64, _i32, _vp, _vp,
                                    _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _f64, _vp, _i32]),
    "msm_sample_transition_matrices": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _f64, C.c_uint64, _i32, _i32, _vp, _i64,
                                              _i32]),
    "msm_reversible_mle": (_i32, [_vp, _vp, _i32, _i32, _f64, _i32, _vp, _i32, _vp, _vp, _vp]),
    "msm_philox4x32": (_i32, [_vp, C.c_uint64, _vp, _vp]),
    "msm_gemm_f64": (_i32, [_vp, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64]),
    "msm_diff_norms": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "msm_gather_f64": (_i32, [_vp, _vp, _i32, _vp, _i64, _vp]),
    "msm_clip_or_wrap": (_i32, [_vp, _vp, _i64, _i64, _f64, _f64, _i32, _vp]),
    "msm_order_statistics": (_i32, [_vp, _vp, _i64, _i64, _vp, _i32, _vp]),
    "msm_weighted_stats": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "msm_hist2d": (_i32, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _f64, _vp, _i32, _vp, _i32, _vp]),
    "msm_smooth_sparse_bins": (_i32, [_vp, _vp, _i32, _i32, _f64, _vp, _vp]),
    "msm_scale_to_total": (_i32, [_vp, _vp, _i32, _f64]),
    "msm_fes_finalize": (_i32, [_vp, _vp, _i32, _f64, _vp, _vp]),
    "msm_kde2d": (_i32, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _f64, _vp, _i32, _vp, _i32, _f64, _f64, _i32, _vp]),
    "msm_solve_f64": (_i32, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "msm_reactive_flux": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "msm_lump_macro": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp]),
    "msm_macro_mfpt": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "msm_silhouette": (_i32, [_vp, _vp, _i64, _i32, _i64, _vp, _i32, _vp, _vp]),
    "msm_grid_cells": (_i32, [_vp, _vp, _i32, _i64, _i32, _i64, _vp, _i32, _vp]),
    "msm_first_occurrence": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "msm_relabel": (_i32, [_vp, _vp, _i64, _vp, _i32, _vp]),
    "msm_ck_test": (_i32, [_vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i32, _vp, _i64, _vp, _vp]),
}

DECLARED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None


def load() -> C.CDLL:
    """Load libmsmhip.so and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m pmarlo_amd.csrc.build` "
            "(hipcc, gfx950).  pmarlo_amd has no CPU fallback."
        )
    handle = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return handle


class _LazyLib:
    def __getattr__(self, name):
        return getattr(load(), name)


lib = _LazyLib()

_EXC = {
    MSM_ERR_INVALID: ValueError,
    MSM_ERR_HIP: MsmError,
    MSM_ERR_NOMEM: MemoryError,
    MSM_ERR_UNSUPPORTED: NotImplementedError,
    MSM_ERR_NOCONV: MsmError,
}


def check(status: int, ctx_handle=None) -> None:
    """Map an msm_status to the Python exception the reference would raise."""
    if status == MSM_OK:
        return
    msg = ""
    if ctx_handle:
        raw = load().msm_last_error(ctx_handle)
        msg = raw.decode("utf-8", "replace") if raw else ""
    exc = _EXC.get(status, MsmError)
    raise exc(msg or f"libmsmhip call failed with status {status}")
