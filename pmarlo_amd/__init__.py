"""MI355X-native MSM-estimation engine behind the pmarlo.features / pmarlo.analysis /
pmarlo.markov_state_model operator API (featurize -> TICA -> k-means -> T-matrix -> ITS).

Host code is Python; all numerics run in hand-written gfx950 HIP kernels reached through the
C ABI of include/msmhip.h (pmarlo_amd/csrc/libmsmhip.so).  There is no CPU fallback."""

__version__ = "0.1.0"
