"""The exchange path of the sharded step under the real RCCL backend.

A one-GPU box cannot host two RCCL ranks, so this runs bench.py as the driver launches it
(torch.distributed.run, backend "nccl") with ONE rank and BENCH_FORCE_EXCHANGE=1: every broadcast,
integer all-reduce, fp64 gather-and-sum and MIN all-reduce of the N > 1 step is issued on the
engine's stream through RCCL.  Over one rank each of them is the identity, so the result must
stay bit-exact against the CPU oracle (the check bench.py itself reports).  World sizes 2 and 4
are covered by the gloo tests in test_dist_gloo.py."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_sharded_step_through_rccl_single_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_FORCE_EXCHANGE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--frames", "200000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["config"]["exchange"] == "rccl"
    par = out["parity"]
    assert "error" not in par, par
    assert par["counts_bit_exact"] and par["labels_bit_exact_given_centres"]
    assert par["its_rel_err"] < 1e-6          # ITS against the numpy oracle on the same counts
    assert par["tica_eig_rel_err"] < 1e-9     # TICA eigenvalues against the oracle on the regenerated shard
