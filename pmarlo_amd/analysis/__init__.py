"""pmarlo.analysis operators on the MI355X engine."""
from .counting import expected_pairs  # noqa: F401
from .discretize import MSMDiscretizationResult, discretize_dataset  # noqa: F401
from .msm import prepare_msm_discretization  # noqa: F401
from .validation import ValidationError, validate_features  # noqa: F401
