"""The exchange path of the sharded step on the GPU box.

(1) RCCL through the C ABI (pmarlo_amd/dist.py NativeComm: msm_comm_init / msm_allreduce_* / msm_broadcast) with ONE
    rank and BENCH_FORCE_EXCHANGE=1: a one-GPU box cannot host two RCCL ranks, so every broadcast, integer
    all-reduce, rank-ordered fp64 sum and MIN all-reduce of the N > 1 step is issued through RCCL over a group of
    one, where each of them is the identity: the result must stay bit-exact against the CPU oracle.
(2) TWO ranks sharing the one GPU with gloo carrying the (device) buffers: the real HIP engine under the real
    ShardedMSM.step() at world = 2.  The all-reduced TICA must equal the oracle's on the LIST of shards, labels the
    oracle's assignment to the shared centres.
bench.py is launched exactly as the driver launches it (torch.distributed.run)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _launch(nproc: int, extra_env: dict, args: list[str]) -> dict:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(nproc), "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs"] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_sharded_step_through_rccl_c_abi_single_rank():
    out = _launch(1, {"BENCH_FORCE_EXCHANGE": "1"}, ["--frames", "200000"])
    assert out["n_gpus"] == 1 and out["config"]["exchange"] == "rccl (C ABI)"
    assert out["config"]["collectives_per_step"] == 13        # 3 + kmeans_iters (moments, start, counts)
    par = out["parity"]
    assert "error" not in par, par
    assert par["counts_bit_exact"] and par["labels_bit_exact_given_centres"]
    assert par["its_rel_err"] < 1e-6          # ITS against the numpy oracle on the same counts
    assert par["tica_eig_rel_err"] < 1e-9     # TICA eigenvalues against the oracle on the regenerated shard


def test_sharded_step_two_ranks_one_gpu_gloo():
    out = _launch(2, {"BENCH_COMM": "torch", "BENCH_BACKEND": "gloo", "BENCH_SAME_GPU": "1"}, ["--frames", "150000"])
    assert out["n_gpus"] == 2 and out["config"]["collectives_per_step"] == 13
    par = out["parity"]
    assert "error" not in par, par
    assert par["labels_bit_exact_given_centres"]
    assert par["tica_eig_rel_err"] < 1e-9 and par["tica_rank"] == par["tica_rank_oracle"]
    assert par["its_rel_err"] < 1e-6


def test_lag_scan_config_through_rccl_c_abi_single_rank():
    """BASELINE config 4 shape: featurize on the device every step, no TICA, 50 lags in one pass and ONE collective."""
    out = _launch(1, {"BENCH_FORCE_EXCHANGE": "1"}, ["--config", "c4", "--frames", "60000"])
    assert out["config"]["collectives_per_step"] == 12        # 2 + kmeans_iters
    par = out["parity"]
    assert "error" not in par, par
    assert par["counts_bit_exact"] and par["labels_bit_exact_given_centres"]
