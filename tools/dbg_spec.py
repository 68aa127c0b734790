import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pmarlo_amd.device import Engine
from pmarlo_amd import _lib
eng = Engine(0)
rng = np.random.default_rng(0)
for n in (6, 40, 300):
    C = rng.random((n, n)) + 5 * np.eye(n)
    T = C / C.sum(1, keepdims=True)
    Td = eng.to_device(T)
    try:
        out = eng.spectrum(Td, n_its=3, allow_unconverged=True, max_launches=2)
        ev = np.linalg.eigvals(T)
        ev = ev[np.argsort(-np.abs(ev))]
        print(n, "ritz", out["ritz"][0][:5], "ref", ev[:5], "res", out["residual"], out["launches"])
    except _lib.MsmError as e:
        print(n, "ERR", e)
