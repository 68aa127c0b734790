"""`python bench.py --gpus N` without a launcher starts N rank processes itself (the driver's call when it does not
wrap the script in torch.distributed.run).  CPU test of the launcher half: the ranks get the environment
torch.distributed.run would give them, join one gloo group from it, rank 0's line is relayed, and a failing rank
makes the whole command fail.  (BENCH_SPAWN_TEST=1 stops the ranks before any engine or GPU work.)"""

from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(n: int, extra: dict):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(BENCH_SPAWN_TEST="1", **extra)
    return subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=300)


def test_gpus_n_starts_n_ranks_with_the_launcher_environment():
    r = _run(2, {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["spawn_test"] and out["n_gpus"] == 2
    assert out["local_ranks_plus_1"] == [1, 2]            # LOCAL_RANK = RANK on one node, every rank present once
    assert out["id_file"] and int(out["master_port"]) > 0  # one RCCL id file and one rendezvous port per launch
    assert not Path(out["id_file"]).parent.exists()        # the launcher cleans up after the ranks


def test_a_failing_rank_fails_the_command():
    r = _run(2, {"BENCH_SPAWN_FAIL_RANK": "1"})
    assert r.returncode != 0
