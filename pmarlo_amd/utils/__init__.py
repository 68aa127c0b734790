"""Helpers around the hot path (mirror of the slice of pmarlo.utils that sits on it)."""
