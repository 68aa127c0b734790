"""pmarlo.markov_state_model operators on the MI355X engine."""
from .ck import CKRunResult, run_ck  # noqa: F401
from .clustering import ClusteringResult, cluster_microstates  # noqa: F401
from .estimation import (build_msm, build_simple_msm, check_transition_matrix, compute_free_energies, count_transitions,  # noqa: F401
                         ensure_connected_counts, finalize_transition_and_stationary, fit_reversible_msm)
from .features import MSMFeatures, ca_distance_pairs, compute_msm_features  # noqa: F401
from .its import (  # noqa: F401
    DEFAULT_ITS_LAGS,
    candidate_lag_ladder,
    compute_implied_timescales,
    detect_timescale_plateau,
    deterministic_its_from_counts,
    safe_timescales,
    select_lag_from_its,
)
from .pcca import canonicalize_macro_labels, pcca_like_macrostates, pcca_memberships  # noqa: F401
from .reduction import pca_reduce, reduce_features, tica_reduce, vamp_reduce  # noqa: F401
from .tpt import (ReactiveFlux, coarse_grain_flux, compute_committor, compute_macro_mfpt,  # noqa: F401
                  compute_macro_populations, find_bottleneck_states, identify_transition_state_ensemble,
                  lump_micro_to_macro_T, pathway_decomposition, reactive_flux)
from .results import ConnectedCountResult, ITSResult, MSMEstimate  # noqa: F401
