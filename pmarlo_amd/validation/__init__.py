"""pmarlo.validation operators on the MI355X engine."""
from .ck_rule import CKConfig, CKDecision, ck_error, decide_ck  # noqa: F401
