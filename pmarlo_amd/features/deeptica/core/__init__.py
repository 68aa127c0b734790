from .trainer_api import _estimate_top_eigenvalues, estimate_top_eigenvalues

__all__ = ["_estimate_top_eigenvalues", "estimate_top_eigenvalues"]
