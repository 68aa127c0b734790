"""pmarlo.api operators on the MI355X engine (the slice the MSM path uses)."""
from .features import compute_features, feature_cache_file, trig_expand_periodic  # noqa: F401
