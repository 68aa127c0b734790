"""pmarlo.features operator API on the MI355X engine (registry, built-ins, featurize)."""
from .base import FEATURE_REGISTRY, FeatureComputer, get_feature, parse_feature_spec, register_feature  # noqa: F401
from .featurize import featurize_trajectory, trig_expand_periodic  # noqa: F401
from . import builtins  # noqa: F401  (registers the built-in featurizers)
