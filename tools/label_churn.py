#!/usr/bin/env python3
"""How many frames change their centre from one Lloyd iteration to the next at the bench shard (C3): the share an
incremental update of the member sums would have to touch."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from pmarlo_amd.device import Engine  # noqa: E402
from pmarlo_amd.dist import ShardConfig, ShardedMSM  # noqa: E402
from tests import _gen  # noqa: E402


def main():
    n, F, d, k, lag = 1_000_000, 64, 10, 500, 10
    eng = Engine(0)
    cfg = ShardConfig(n_frames=n, n_features=F, tica_dim=d, k=k, lag=lag, kmeans_iters=0, seed=0, n_total=n)
    msm = ShardedMSM(eng, cfg, eng.to_device(_gen.correlated_series(n, F, seed=1000)))
    msm.step()                       # TICA + projection + seeded centres, no Lloyd iteration
    b = msm.buf
    prev = None
    for it in range(12):
        lab = eng.kmeans_assign(msm.Y, b["centers"]).to_host()
        if prev is not None:
            print(f"iteration {it}: {np.mean(lab != prev) * 100:.2f} % of the frames changed centre")
        prev = lab
        eng.kmeans_accumulate(msm.Y, b["centers"], b["fit_state"], msm.km_sums, msm.km_counts)
        eng.kmeans_update(msm.km_sums, msm.km_counts, b["centers"], b["fit_state"], clear=True)


if __name__ == "__main__":
    main()
