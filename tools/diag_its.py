import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pmarlo_amd.device import get_engine
from tools.time_its import chain
from pmarlo_amd._lib import check, lib

k, lag = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = get_engine()
x = np.concatenate([chain(k, 60_000, s) for s in range(4)])
C = np.zeros((k, k), np.int64)
np.add.at(C, (x[:-lag], x[lag:]), 1)
cd = eng.to_device(C)
T, act, inv = eng.empty((k, k), np.float64), eng.empty((k,), np.int32), eng.empty((k,), np.int32)
na, rows = eng.empty((1,), np.int32), eng.empty((k,), np.float64)
check(lib.msm_transition_matrix(eng.handle, cd.ptr, 0, k, 1, 1e-3, 1e-12, T.ptr, act.ptr, inv.ptr, na.ptr, rows.ptr, None), eng.handle)
ev = np.linalg.eigvals(T.to_host())
ev = ev[np.argsort(-np.abs(ev))]
print("lag", lag, "|eig| top 10:", np.round(np.abs(ev[:10]), 4), " |eig| 33:", round(abs(ev[32]), 4))
for S in (1, 200):
    Td = eng.sample_transition_matrices(cd, act, na, alpha=1e-3, seed=5, n_samples=S) if S > 1 else T.view((1, k, k))
    for n_its in (3, 5):
        eng.sync(); t0 = time.perf_counter()
        spec = eng.spectrum(Td, n=eng.to_device(np.full(S, k, np.int32)), n_its=n_its, lags=np.full(S, float(lag)), want_pi=False, allow_unconverged=True)
        eng.sync(); dt = time.perf_counter() - t0
        print(f"S={S} n_its={n_its}: {dt*1e3:.1f} ms launches={spec['launches']} p={spec['p']} worst residual={spec['residual'].max():.2e} frac>1e-9: {(spec['residual']>1e-9).mean():.3f}")
