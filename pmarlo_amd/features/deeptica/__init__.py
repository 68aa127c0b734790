"""DeepTICA helpers that sit on the TICA path (mirror of pmarlo.features.deeptica): only the
numpy-level eigenvalue estimator of the trainer is built here; the neural CV model itself stays
PyTorch in the reference and is out of scope."""
