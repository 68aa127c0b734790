#!/usr/bin/env python3
"""Benchmark of the hot path: (featurize ->) TICA -> k-means -> transition matrix on one shard per GPU.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched by
torch.distributed.run with one rank per GPU (only RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT are read: the
exchange is RCCL through the C ABI, pmarlo_amd/dist.py NativeComm).  Rank 0 prints ONE JSON line.
Started WITHOUT a launcher (no RANK in the environment) and with --gpus N > 1, this process only starts N rank
processes of itself (it never touches a GPU), relays rank 0's line and exits non-zero if any rank failed.

Headline workload (BASELINE.json metric "... 1Mx64 synth", --config c3): per GPU one synthetic shard of
1,000,000 frames x 64 float32 features (AR(1)-latent generator of the reference's
tests/perf/test_tica_perf.py:65-81, vectorised; seed 1000 + rank), resident in HBM when the timed region starts.
One step = one pass of the whole path over the shard: fp64-MFMA lagged covariance (the standardisation sums
come out of it) -> on-device TICA solve (dim 10) -> projection (with max |Y|) -> bf16 frame images -> k-means
(k = 500, seeded start + 10 full-batch Lloyd iterations: certified bf16 matrix-core filter + pinned fp64
refinement) -> final assignment -> lag-10 transition counts -> row-normalised T.
Weak scaling: every rank holds its own shard; only the small moment / member-sum / count buffers are all-reduced.

Other workloads (--config): c4 = BASELINE config 4 (chignolin: xyz -> 45 C-alpha distances on the device every
step -> k = 200 in the feature space -> lag scan 1..50 in one pass and one collective); c5 / c5t = BASELINE
config 5 per GPU (1.25 M x 256 float32, k = 2000; clustering in the raw 256-d space / after TICA -> 10).
Extra legs at N = 1 (reported beside the headline, never as it): the featurizer alone and in front of the c3
chain, the operator API from host arrays, k-means quality against the CPU baseline, per-stage times.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_MFMA_PEAK_TF = 78.6       # MI355X fp64 matrix (= vector) peak
BF16_MFMA_PEAK_TF = 2500.0     # dense bf16 matrix peak

CONFIGS = {
    #        frames/GPU  F    tica  k     lag  iters
    "c3": dict(n=1_000_000, F=64, d=10, k=500, lag=10, iters=10),
    "c4": dict(n=1_000_000, F=45, d=0, k=200, lag=10, iters=10, lags=tuple(range(1, 51)), atoms=138),
    "c5": dict(n=1_250_000, F=256, d=0, k=2000, lag=10, iters=10),
    "c5t": dict(n=1_250_000, F=256, d=10, k=2000, lag=10, iters=10),
}


def chignolin_frames(n: int, seed: int):
    """BASELINE config 4 input: the 18 NMR models of data/chignolin.pdb tiled cyclically (nm),
    plus N(0, 0.02 nm) noise; coordinates and C-alpha pairs come from the committed fixture."""
    g = np.load(ROOT / "tests" / "golden" / "featurizer.npz")
    base = g["chig_xyz"][:18].astype(np.float32)
    rng = np.random.default_rng(seed)
    xyz = base[np.arange(n) % 18] + rng.normal(0.0, 0.02, size=(n, base.shape[1], 3)).astype(np.float32)
    return xyz, g["chig_pairs"].astype(np.int32)


# the reference's own operators at this size, timed in the survey container (8 vCPU)
REFERENCE_FUNCTIONS_FRAMES_PER_S = 44_000.0


def cpu_baseline(X: np.ndarray, c: dict, reps: int = 5) -> dict:
    """The oracle's numpy / scikit-learn restatement of the same path (what the reference runs at this size:
    _preprocess -> TICA -> MiniBatchKMeans branch (n*d >= 5e6) -> predict -> _weighted_counts -> _normalise_counts).
    SURVEY section 8d protocol: one warm-up run, then the median of `reps` runs, on the host's thread pools as they are."""
    from oracle import npport

    def once():
        t0 = time.perf_counter()
        Xp = npport.preprocess(X, scale=True)
        model = npport.tica_fit([Xp], c["lag"], dim=c["d"])
        Y = npport.tica_transform(model, Xp)
        fit = npport.kmeans_discretizer_fit(Y, c["k"], random_state=0)
        Yz = (Y - fit["mean"]) / fit["std_safe"]
        labels = npport.kmeans_predict(Yz, fit["centers"])
        counts, _ = npport.weighted_counts(labels, c["k"], c["lag"])
        npport.normalise_counts(counts)
        return time.perf_counter() - t0, model, Yz, fit, labels

    once()                                   # warm-up (page faults, BLAS thread pools, sklearn imports)
    runs = [once() for _ in range(max(1, reps))]
    times = sorted(r[0] for r in runs)
    dt = float(np.median(times))
    _, model, Yz, fit, labels = runs[-1]
    inertia = float(((Yz - fit["centers"][labels]) ** 2).sum())
    try:
        from threadpoolctl import threadpool_info

        pools = [{k: p.get(k) for k in ("user_api", "internal_api", "num_threads")} for p in threadpool_info()]
        threads = max([p.get("num_threads") or 1 for p in pools] + [1])
    except Exception:
        pools, threads = [], os.cpu_count() or 1
    v = X.shape[0] / dt
    return {"value": v, "unit": "frames/s", "cores": int(threads), "kind": "port",
            "sample": f"the full {X.shape[0]}x{X.shape[1]} shard: 1 warm-up + {len(times)} runs, median {dt:.2f} s "
                      f"(min {times[0]:.2f}, max {times[-1]:.2f}; numpy TICA + sklearn MiniBatchKMeans + predict + numpy counts)",
            "runs_s": times, "host_cpus": os.cpu_count(), "threadpool_info": pools,
            "note": f"{v / REFERENCE_FUNCTIONS_FRAMES_PER_S:.1f}x the survey's timing of the reference's own functions at this "
                    f"size (BASELINE.md section 2: ~{REFERENCE_FUNCTIONS_FRAMES_PER_S:.0f} frames/s on 8 vCPU, dominated by "
                    "_preprocess): the port is the faster stand-in, so the GPU / CPU ratio is understated, not inflated",
            "tica_eigenvalues": model["eigenvalues"][:c["d"]].tolist(), "inertia_whitened": inertia,
            "_Yz": Yz}


def kernel_rooflines(eng, msm, c: dict, n: int, acc_ms: float) -> list:
    """The kernels of the step, each launched alone between HIP events on the engine's stream (median of 10), with
    the algorithmic work of SURVEY section 8d, the work the kernel really executes, and the fraction of the roof it runs
    under.  profiles/r03_bench_kernel_stats.md holds rocprofv3's durations of the same kernels inside the step."""
    F, d, k, lag = c["F"], c["d"], c["k"], c["lag"]
    b = msm.buf

    def med(fn, reps=10):
        for _ in range(2):
            fn()
        eng.sync()
        evs = []
        for _ in range(reps):
            a, e = eng.event(), eng.event()
            a.record()
            fn()
            e.record()
            evs.append((a, e))
        eng.sync()
        return float(np.median([a.elapsed_ms(e) for a, e in evs]))

    out = []
    s = 4   # bytes per element of X (float32 shard)

    def row(name, ms, bound, alg_flops=None, exe_flops=None, alg_bytes=None, peak=None, note=None):
        r = {"kernel": name, "avg_us": ms * 1e3, "bound": bound}
        if alg_flops is not None:
            r["algorithmic_flops"] = alg_flops
        if exe_flops is not None:
            r["executed_flops"] = exe_flops
        if alg_bytes is not None:
            r["algorithmic_bytes"] = alg_bytes
        if bound == "mfma" and peak:
            r.update(achieved=(exe_flops or alg_flops) / (ms * 1e-3) / 1e12, peak=peak, unit="TFLOP/s")
            r["frac"] = r["achieved"] / peak
            if alg_flops is not None:
                r["frac_algorithmic"] = alg_flops / (ms * 1e-3) / 1e12 / peak
        elif bound == "hbm":
            r.update(achieved=alg_bytes / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
            r["frac"] = r["achieved"] / HBM_PEAK_GBS
        if note:
            r["note"] = note
        out.append(r)

    if d:
        ms = med(lambda: eng.lagged_moments(msm.x, lag, b["shift"], assume_finite=True, out=b["lagged"], symmetric=True))
        row("cov_fused_kernel (+ cov_reduce_kernel), symmetric flavour", ms, "mfma", alg_flops=3.0 * F * F * n,
            exe_flops=(2.0 * F * F + 32.0 * F) * n, alg_bytes=float(F * s * n), peak=FP64_MFMA_PEAK_TF,
            note="SURVEY 8d prices C(0) + C(tau) at 3 F^2 flop per frame; the reversible estimator accumulates "
                 "S = sum (x + y)(x + y)' and M00 only: 2 F^2 + 32 F executed")
        ms = med(lambda: eng.tica_solve(b["lagged"], F, scale=msm.scale, epsilon=1e-6, kinetic_map=True,
                                        out=(msm.eig, msm.W, msm.m2, msm.rank_d)))
        row("tica_solve_kernel (one workgroup)", ms, "latency", alg_flops=10.0 * F ** 3,
            note="F x F generalised symmetric eigenproblem on one CU: no roof applies (SURVEY 8d: latency-bound)")
        ms = med(lambda: eng.project(msm.x, b["shift"], msm.inv_scale, msm.W, d, mean2=msm.m2, out=msm.Y,
                                     absmax=b["fit_state"].view((1,), offset_elems=2), assume_finite=True))
        row("project_mfma_kernel", ms, "hbm", alg_flops=2.0 * F * d * n, alg_bytes=float((F * s + d * 8) * n))
    dc = msm.cfg.cluster_dim
    if msm.km_image is not None:
        ms = med(lambda: eng.kmeans_pack(msm.Y, image=msm.km_image))
        img_b = eng.kmeans_image_bytes(n, dc) / n
        row("kmeans_pack_kernel (frame images, once per step)", ms, "hbm",
            alg_bytes=float((dc * msm.Y.dtype.itemsize + img_b) * n))
        nm = 1 if 6 * dc + 4 <= 32 else 2
        tiles = ((((k + 15) // 16) + 1) // 2) * 2
        exe = float(-(-n // 16)) * tiles * nm * 16384.0
        row("kmeans_filter_kernel assign + accumulate (x kmeans_iters per step; time = mean over the timed steps)", acc_ms,
            "mfma", alg_flops=2.0 * k * dc * n, exe_flops=exe, alg_bytes=float((img_b + dc * msm.Y.dtype.itemsize + 8) * n),
            peak=BF16_MFMA_PEAK_TF, note="executed = bf16 16x16x32 instructions issued x 16384 flop")
        lab = eng.empty((n,), np.int32)
        ms = med(lambda: eng.kmeans_assign(msm.Y, b["centers"], labels=lab, image=msm.km_image))
        row("kmeans_filter_kernel assign (final labels)", ms, "mfma", alg_flops=2.0 * k * dc * n, exe_flops=exe,
            alg_bytes=float((img_b + dc * msm.Y.dtype.itemsize + 4) * n), peak=BF16_MFMA_PEAK_TF)
    if not msm.cfg.lags:
        ms = med(lambda: eng.count_transitions(msm.labels, k, lag, out=b["counts"].view((k, k)),
                                               pairs=b["counts"].view((1,), offset_elems=k * k)))
        row("count_bucket_scatter_kernel + count_bucket_bin_kernel", ms, "hbm", alg_bytes=float(4 * n + 8 * k * k),
            note="4 B label per frame + the int64 matrix once; two launches, no global atomics: latency of the two "
                 "short launches, not bandwidth, sets the pace")
    return out


def featurize_legs(eng, c3: dict) -> dict:
    """The featurizer alone (HBM GB/s against the algorithmic 12 A + 4 F bytes per frame) at the C4 shape and at a
    64-feature synthetic shape, and the c3 chain with the featurizer in front (xyz resident, features rebuilt by
    every step)."""
    from pmarlo_amd.dist import ShardConfig, ShardedMSM

    out = {}
    n = 1_000_000
    for name, A, F in (("c4_chignolin_45_ca_pairs", 138, 45), ("synthetic_128_atoms_64_pairs", 128, 64)):
        rng = np.random.default_rng(7)
        if A == 138:
            xyz, pairs = chignolin_frames(n, 1234)
        else:
            xyz = rng.normal(0.0, 1.0, size=(n, A, 3)).astype(np.float32)
            pairs = np.stack([rng.integers(0, A, F), rng.integers(0, A, F)], 1).astype(np.int32)
            pairs[:, 1] = np.where(pairs[:, 1] == pairs[:, 0], (pairs[:, 0] + 1) % A, pairs[:, 1])
        xd = eng.to_device(xyz)
        pd = eng.to_device(pairs)
        feat = eng.empty((n, F), np.float32)
        for _ in range(3):
            eng.featurize_distances_into(xd, pd, feat)
        eng.sync()
        evs = []
        for _ in range(10):
            a, b = eng.event(), eng.event()
            a.record()
            eng.featurize_distances_into(xd, pd, feat)
            b.record()
            evs.append((a, b))
        eng.sync()
        ms = float(np.median([a.elapsed_ms(b) for a, b in evs]))
        bytes_alg = n * (12 * A + 4 * F)
        out[name] = {"frames": n, "atoms": A, "features": F, "ms": ms, "frames_per_s": n / (ms * 1e-3),
                     "algorithmic_bytes": bytes_alg, "GBps": bytes_alg / (ms * 1e-3) / 1e9,
                     "frac_of_hbm_peak": bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if A == 128:
            # the headline chain behind the featurizer: xyz -> 64 distances -> TICA -> k-means -> T
            cfg = ShardConfig(n_frames=n, n_features=F, tica_dim=c3["d"], k=c3["k"], lag=c3["lag"],
                              kmeans_iters=c3["iters"], seed=0, n_total=n, n_atoms=A, pairs=pairs)
            msm = ShardedMSM(eng, cfg, xd)
            for _ in range(2):
                msm.step()
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                msm.step()
            eng.sync()
            dt = (time.perf_counter() - t0) / 5
            out["chain_with_featurizer"] = {"workload": f"xyz {n} x {A} atoms -> {F} pair distances -> the c3 chain",
                                            "ms_per_step": dt * 1e3, "frames_per_s": n / dt}
            del msm
        del xd, feat
    return out


def operator_api_leg(X: np.ndarray, c: dict) -> dict:
    """The pmarlo-shaped operators from HOST arrays (upload and download inside the clock): the first call of the
    process (allocations, first touch of fresh host memory) and the median of three calls after it."""
    from pmarlo_amd.analysis.discretize import discretize_dataset
    from pmarlo_amd.markov_state_model.clustering import cluster_microstates
    from pmarlo_amd.markov_state_model.reduction import tica_reduce

    def timed(fn, reps=3):
        t0 = time.perf_counter()
        res = fn()
        first = time.perf_counter() - t0
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            res = fn()
            ts.append(time.perf_counter() - t0)
        return res, first * 1e3, float(np.median(ts)) * 1e3

    out = {}
    Y, f1, m1 = timed(lambda: tica_reduce(X, lag=c["lag"], n_components=c["d"]))
    res, f2, m2 = timed(lambda: cluster_microstates(Y, method="kmeans", n_states=c["k"], random_state=0, max_iter=c["iters"],
                                                    tolerance=0.0))
    ds = {"splits": {"train": {"X": Y, "segments": [{"start": 0, "stop": Y.shape[0]}]}}}
    dres, f3, m3 = timed(lambda: discretize_dataset(ds, cluster_mode="kmeans", n_microstates=c["k"], lag_time=c["lag"],
                                                    random_state=0))
    n = X.shape[0]
    out["tica_reduce_ms"], out["cluster_microstates_ms"], out["discretize_dataset_ms"] = m1, m2, m3
    out["first_call_ms"] = {"tica_reduce": f1, "cluster_microstates": f2, "discretize_dataset": f3}
    out["frames_per_s_tica_plus_cluster"] = n / ((m1 + m2) * 1e-3)
    out["frames_per_s_all_three"] = n / ((m1 + m2 + m3) * 1e-3)
    out["pcie_floor_ms"] = {"tica_reduce": (X.nbytes + Y.nbytes) / 55e9 * 1e3, "cluster_microstates": Y.nbytes / 55e9 * 1e3,
                            "discretize_dataset": Y.nbytes / 55e9 * 1e3}
    out["n_states_found"] = int(res.n_states)
    out["counted_pairs"] = int(dres.counted_pairs.get("train", 0)) if hasattr(dres, "counted_pairs") else None
    out["note"] = ("host numpy in, host numpy out: PCIe transfers (pageable memory, ~55 GB/s up; pcie_floor_ms = bytes moved "
                   "at that rate) and host bookkeeping are inside these times; *_ms = median of three calls after the first")
    return out


def spawn_ranks(n_ranks: int, argv: list[str]) -> int:
    """`bench.py --gpus N` without a launcher: start N rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* as torch.distributed.run sets them, one GPU each), relay rank 0's stdout, return non-zero if any rank
    failed.  This process never imports the engine or torch: it must not touch a GPU it does not use."""
    import socket
    import subprocess
    import tempfile

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    id_dir = tempfile.mkdtemp(prefix="msm_comm_")
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MSM_COMM_ID_FILE=os.path.join(id_dir, "rccl.id"),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env, cwd=str(ROOT),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    deadline = time.monotonic() + 120.0
    for pr in procs[1:]:
        try:
            pr.wait(timeout=max(1.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            pr.kill()           # the exact child this process started
            pr.wait()
        rc = rc or pr.returncode
    try:
        for f in os.listdir(id_dir):
            os.unlink(os.path.join(id_dir, f))
        os.rmdir(id_dir)
    except OSError:
        pass
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    return int(rc != 0)


def spawn_test_rank() -> None:
    """BENCH_SPAWN_TEST=1 (tests/test_bench_spawn.py, CPU): a rank started by `spawn_ranks` joins a gloo group built
    from the environment it was given and rank 0 reports what every rank saw.  No engine, no GPU, no numbers."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = torch.zeros(world, dtype=torch.int64)
    seen[rank] = 1 + int(os.environ["LOCAL_RANK"])
    dist.all_reduce(seen)
    if os.environ.get("BENCH_SPAWN_FAIL_RANK") == str(rank):
        dist.destroy_process_group()
        sys.exit(3)
    if rank == 0:
        print(json.dumps({"spawn_test": True, "n_gpus": dist.get_world_size(), "local_ranks_plus_1": seen.tolist(),
                          "master_port": os.environ["MASTER_PORT"], "id_file": os.environ.get("MSM_COMM_ID_FILE")}), flush=True)
    dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the featurizer / operator-API / quality legs")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (default: the config's)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("BENCH_SPAWN_TEST") == "1":
        spawn_test_rank()
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_FORCE_EXCHANGE=1 (one rank): keep every collective in the step, so a one-GPU box exercises the
    # RCCL calls the N > 1 runs make
    multi = world > 1 or os.environ.get("BENCH_FORCE_EXCHANGE") == "1"
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from pmarlo_amd.device import Engine
    from pmarlo_amd.dist import (NativeComm, ShardConfig, ShardedMSM, TorchComm, exchange_aliases, exchange_shapes,
                                 torch_exchange_buffers)
    from tests import _gen

    c = dict(CONFIGS[args.config])
    n = int(args.frames) if args.frames else c["n"]
    F, d, k, lag, iters = c["F"], c["d"], c["k"], c["lag"], c["iters"]
    cfg_kw = dict(n_frames=n, n_features=F, tica_dim=d, k=k, lag=lag, kmeans_iters=iters, seed=0, n_total=n * world,
                  lags=c.get("lags"))

    # ---- transport: RCCL through the C ABI (default), or torch.distributed (BENCH_COMM=torch; BENCH_BACKEND=gloo
    # and BENCH_SAME_GPU=1 rehearse several ranks on one GPU)
    comm_kind = os.environ.get("BENCH_COMM", "rccl") if multi else None
    comm, shared, dist = None, None, None
    same_gpu = os.environ.get("BENCH_SAME_GPU") == "1"
    if same_gpu:
        local_rank = 0
    if comm_kind == "torch":
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("BENCH_BACKEND", "nccl"))
        stream = torch.cuda.Stream(device=local_rank)
        torch.cuda.set_stream(stream)
        eng = Engine(local_rank, stream=stream.cuda_stream)
    else:
        eng = Engine(local_rank)

    if c.get("atoms"):
        xyz, pairs = chignolin_frames(n, 1234 + rank)
        X = None
        xd = eng.to_device(xyz)
        cfg = ShardConfig(**cfg_kw, n_atoms=c["atoms"], pairs=pairs)
    else:
        X = _gen.correlated_series(n, F, seed=1000 + rank)
        xd = eng.to_device(X)
        cfg = ShardConfig(**cfg_kw)

    if comm_kind == "torch":
        tensors, shared = torch_exchange_buffers(eng, cfg, torch.device("cuda", local_rank))
        tensors["timing"] = torch.zeros((1,), dtype=torch.float64, device=torch.device("cuda", local_rank))
        shared["timing"] = eng.wrap(tensors["timing"].data_ptr(), (1,), np.dtype(np.float64))
        comm = TorchComm(tensors, shared)
    elif comm_kind == "rccl":
        shared = {nm: eng.zeros(shape, np.dtype(dt)) for nm, (shape, dt) in exchange_shapes(cfg).items()}
        shared["timing"] = eng.zeros((1,), np.float64)
        comm = NativeComm.from_env(eng, shared)
    msm = ShardedMSM(eng, cfg, xd, comm=comm, shared=shared, always_exchange=multi)

    def barrier():
        eng.sync()
        if multi:
            comm.allreduce_max("timing")     # a collective every rank must enter
            eng.sync()
            if comm_kind == "torch":
                torch.cuda.synchronize()

    for _ in range(max(0, args.warmup)):
        msm.step()
    barrier()
    msm.time_accum = True
    msm.time_stages = True
    if multi and hasattr(comm, "events"):
        comm.events.clear()
        comm.timing = True
    n_coll0 = comm.n_collectives if multi else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        msm.step()
    barrier()
    elapsed = time.perf_counter() - t0
    msm.time_accum = False
    msm.time_stages = False
    exchange_ms = None
    if multi and hasattr(comm, "events"):
        comm.timing = False
        # device time between HIP events around the collectives of each exchange buffer, per step (the barrier's own
        # "timing" collective is outside the timed steps)
        exchange_ms = {nm: v / args.steps for nm, v in comm.exchange_ms().items() if nm != "timing"}
        exchange_ms["total"] = float(sum(exchange_ms.values()))
    n_coll = (comm.n_collectives - n_coll0 - 1) if multi else 0     # minus the barrier's own collective
    n_gpus = comm.comm_info()[1] if (multi and hasattr(comm, "comm_info")) else (comm.world if multi else 1)
    if multi:
        shared["timing"].copy_from_host(np.array([elapsed]))
        comm.allreduce_max("timing")
        eng.sync()
        elapsed = float(shared["timing"].to_host()[0])

    # ---- per-stage device times (HIP events on the engine's stream, inside the timed steps)
    names = [s for s in ("begin", "featurize", "moments", "tica_solve", "project", "kmeans", "counts") if s in msm.stage_events]
    stages = {}
    for a, b_ in zip(names[:-1], names[1:]):
        stages[b_] = float(np.mean([x.elapsed_ms(y) for x, y in zip(msm.stage_events[a], msm.stage_events[b_])]))

    # dominant kernel: the k-means assign + accumulate pass, measured with HIP events on the engine's stream
    acc_ms = [a.elapsed_ms(b) for a, b in msm.accum_events]
    acc_ms_avg = float(np.mean(acc_ms)) if acc_ms else float("nan")
    dc = cfg.cluster_dim
    flops_per_launch = 2.0 * k * dc * n            # 2 k d per frame
    filtered = msm.km_image is not None
    if filtered:
        # the pass issues bf16 matrix instructions (an exact filter, labels bit-identical to the fp64 chain): the roof it
        # runs under is the bf16 matrix peak for the instructions it executes, with the HBM fraction beside it
        nm = 1 if 6 * dc + 4 <= 32 else 2
        tiles = ((((k + 15) // 16) + 1) // 2) * 2
        exec_flops = float(-(-n // 16)) * tiles * nm * 16384.0      # bf16 16x16x32 instructions issued x 16384 flop
        img_b = eng.kmeans_image_bytes(n, dc) / n
        alg_bytes = float((img_b + dc * msm.Y.dtype.itemsize + 8) * n)    # frame images + coordinates + the label in and out
        achieved_tf = exec_flops / (acc_ms_avg * 1e-3) / 1e12
        roofline = {
            "bound": "mfma", "achieved": achieved_tf, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": achieved_tf / BF16_MFMA_PEAK_TF, "traffic": None, "dtype_executed": "bf16",
            "kernel": "kmeans_filter_kernel<double,2,4,true,false> (assign + accumulate: bf16x3 matrix-core filter, fp32 "
                      "candidate pick, fp64 only for frames without a certificate)",
            "launch_ms": acc_ms_avg, "launches_timed": len(acc_ms), "executed_flops_per_launch": exec_flops,
            "algorithmic_flops_per_launch": flops_per_launch, "algorithmic_bytes_per_launch": alg_bytes,
            "hbm": {"achieved": alg_bytes / (acc_ms_avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": alg_bytes / (acc_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "speedup_vs_fp64_bound": flops_per_launch / (acc_ms_avg * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF,
            "note": "achieved / peak / frac: bf16 matrix work EXECUTED per launch / launch time against the dense bf16 peak. "
                    "speedup_vs_fp64_bound: the algorithmic 2 k d flop per frame / launch time over the fp64 matrix peak an "
                    "all-fp64 pass is bounded by -- what the exact filter buys, not a fraction of any roof."}
    else:
        achieved_tf = flops_per_launch / (acc_ms_avg * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": achieved_tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": achieved_tf / FP64_MFMA_PEAK_TF, "traffic": None,
                    "kernel": "kmeans_mfma_kernel (assign + accumulate, fp64 MFMA)", "launch_ms": acc_ms_avg,
                    "launches_timed": len(acc_ms), "algorithmic_flops_per_launch": flops_per_launch}
    prof = ROOT / "profiles" / "r03_pmc_hbm.json"
    if prof.exists() and args.config == "c3" and n == CONFIGS["c3"]["n"]:
        try:
            pj = json.loads(prof.read_text())
            if pj.get("kernel_prefix", "x") in roofline["kernel"]:
                roofline["traffic"] = pj.get("bytes_per_launch")
                roofline["traffic_unit"] = ("bytes per launch, rocprofv3 PMC of an earlier run of this command kept in "
                                            "profiles/r03_pmc_hbm.md (not re-measured by this run)")
        except Exception:
            pass

    if rank == 0:
        total_frames = float(n) * world * args.steps
        workload = (f"synthetic {n} frames x {F} f32 features per GPU (AR(1) latents), "
                    + (f"TICA->{d}, " if d else "no TICA, ") + f"k={k} ({iters} Lloyd iterations), lag={lag}, row-normalised T")
        if c.get("atoms"):
            workload = (f"chignolin-shaped xyz {n} frames x {c['atoms']} atoms per GPU -> {F} C-alpha distances on the device, "
                        f"k={k} in the feature space ({iters} Lloyd iterations), lag scan 1..{len(c['lags'])} in one pass")
        out = {
            "metric": "frames/sec featurize->TICA->k-means->T-matrix, 1Mx64 synth; ITS rel-err",
            "value": total_frames / elapsed,
            "unit": "frames/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "frames_per_gpu": n, "features": F, "tica_dim": d, "k": k,
                       "lag": lag, "kmeans_iters": iters, "parallelism": f"shards{world}",
                       "exchange": ({"rccl": "rccl (C ABI)", "torch": "torch.distributed " + os.environ.get("BENCH_BACKEND", "nccl")}
                                    [comm_kind] if multi else None),
                       "collectives_per_step": (n_coll / args.steps if multi else 0)},
            "roofline": roofline,
            "stages_ms": stages,
        }
        if exchange_ms is not None:
            out["exchange_ms"] = exchange_ms
            out["config"]["exchange_note"] = ("RCCL defaults (no scaling curve has been measured for this path yet: the pool "
                                              "offers one GPU per box)")
        # ---- parity of this very run against the CPU oracle (outside the timed region) ----
        parity = {}
        try:
            from oracle import cport, npport

            L = len(c["lags"]) if c.get("lags") else 1
            li = list(c["lags"]).index(lag) if c.get("lags") and lag in c["lags"] else 0
            counts_all = msm.buf["counts"].view((L, k, k)).to_host()
            pairs_all = msm.buf["counts"].view((L,), offset_elems=L * k * k).to_host()
            counts = counts_all[li]
            labels = msm.labels.to_host()
            centers = msm.buf["centers"].to_host()
            nchk = min(n, 250_000)      # the oracle's assignment is a scalar loop: a quarter of a million frames
            Yh = (msm.Y.view((nchk, dc)) if True else msm.Y).to_host().astype(np.float64)
            parity["labels_bit_exact_given_centres"] = bool(np.array_equal(labels[:nchk], cport.kmeans_assign(Yh, centers)))
            parity["labels_checked"] = nchk
            if world == 1:
                ok = True
                for i in range(L):
                    want, pw = cport.count_transitions(labels, k, int(c["lags"][i]) if c.get("lags") else lag)
                    ok = ok and bool(np.array_equal(counts_all[i], want)) and int(pairs_all[i]) == pw
                parity["counts_bit_exact"] = ok
            tm = eng.transition_matrix(eng.to_device(counts), mode=1)
            spec = eng.spectrum(tm["T"], n=tm["n_active"], n_its=5, lags=[float(c["lags"][li] if c.get("lags") else lag)],
                                allow_unconverged=True)
            ev_ref, ts_ref = npport.its_from_counts(counts, int(c["lags"][li] if c.get("lags") else lag), 5)
            okm = np.isfinite(ts_ref)
            parity["its_rel_err"] = float(np.max(np.abs(spec["its_ts"][0][okm] - ts_ref[okm]) / ts_ref[okm]))
            parity["its_timescales_frames"] = spec["its_ts"][0].tolist()
            parity["its_residual"] = float(spec["residual"][0])
            if d:
                parity["tica_eigenvalues"] = msm.eig.to_host()[:d].tolist()
                parity["tica_rank"] = int(msm.rank_d.to_host()[0])
            if filtered:
                parity["kmeans_filter_scanned_frames_per_pass"] = eng.kmeans_filter_scanned() / max(
                    1, (args.steps + args.warmup) * (iters + 1))
            if d and multi and n * world <= 2_000_000:
                # rank 0 regenerates every shard (seeded) and checks the all-reduced TICA against the
                # oracle on the list of shards (pairs never cross a shard)
                allx = npport.preprocess(np.vstack([_gen.correlated_series(n, F, seed=1000 + r) for r in range(world)]),
                                         scale=True)
                ref = npport.tica_fit([allx[r * n:(r + 1) * n] for r in range(world)], lag, dim=d)
                got = np.asarray(parity["tica_eigenvalues"])
                m = min(2 * world, d)  # the resolved slow modes (2 per independently mixed shard)
                parity["tica_eig_rel_err"] = float(np.max(np.abs(got[:m] - ref["eigenvalues"][:m])
                                                          / np.abs(ref["eigenvalues"][:m])))
                parity["tica_rank_oracle"] = int(ref["rank"])
        except Exception as exc:  # parity reporting must not hide the throughput line
            parity["error"] = repr(exc)
        out["parity"] = parity

        extra = (not multi) and (not args.no_extra_legs) and args.config == "c3" and X is not None
        # from-host pass: the same step with the shard uploaded from pageable host
        # memory inside the clock -- reported beside `value`, never as it
        if not multi and X is not None:
            eng.sync()
            t1 = time.perf_counter()
            xd2 = eng.to_device(X)
            eng.sync()
            h2d = time.perf_counter() - t1
            msm.x = xd2
            msm.step()
            eng.sync()
            tot = time.perf_counter() - t1
            msm.x = xd
            del xd2
            out["from_host"] = {"value": n / tot, "unit": "frames/s", "h2d_ms": h2d * 1e3, "step_ms": (tot - h2d) * 1e3,
                                "h2d_GBps": X.nbytes / h2d / 1e9, "note": "one pageable-memory upload + one step"}
        if not multi and not args.no_extra_legs and X is not None:
            try:
                roofline["kernels"] = kernel_rooflines(eng, msm, c, n, acc_ms_avg)
            except Exception as exc:
                roofline["kernels"] = [{"error": repr(exc)}]
        if not multi and not args.no_cpu_baseline and X is not None and d:
            cb = cpu_baseline(X, c)
            ref_eig = np.asarray(cb.pop("tica_eigenvalues"))
            Yz = cb.pop("_Yz")
            got = np.asarray(parity.get("tica_eigenvalues", ref_eig))
            out["parity"]["tica_eig_rel_err"] = float(np.max(np.abs(got - ref_eig) / np.abs(ref_eig)))
            out["cpu_baseline"] = cb
            from threadpoolctl import threadpool_limits

            with threadpool_limits(limits=1):
                cb1 = cpu_baseline(X, c)
            for key in ("tica_eigenvalues", "_Yz", "threadpool_info"):
                cb1.pop(key, None)
            cb1["cores"] = 1
            out["cpu_baseline_1thread"] = cb1
            # what the fast k-means buys: the device's 10 Lloyd iterations and the CPU branch (MiniBatchKMeans) on
            # the SAME whitened coordinates, inertia = sum of squared distances to the assigned centre
            try:
                yz = eng.to_device(np.ascontiguousarray(Yz))
                cen, st = eng.kmeans_fit(yz, k, seed=0, max_iter=iters, tol2=0.0)
                md = eng.empty((n,), np.float64)
                eng.kmeans_assign(yz, cen, mindist=md)
                dev_inertia = float(eng.sum_f64(md).to_host()[0])
                cen100, _ = eng.kmeans_fit(yz, k, seed=0, max_iter=100, tol2=1e-4 * dc)
                eng.kmeans_assign(yz, cen100, mindist=md)
                out["kmeans_quality"] = {
                    "space": "TICA coordinates whitened as the reference's discretizer does (z-score, ddof = 1)",
                    "inertia_device_10_lloyd": dev_inertia,
                    "inertia_device_to_tolerance": float(eng.sum_f64(md).to_host()[0]),
                    "inertia_cpu_baseline_minibatch": cb["inertia_whitened"],
                    "ratio_device_over_cpu": dev_inertia / cb["inertia_whitened"]}
                del yz, md
            except Exception as exc:
                out["kmeans_quality"] = {"error": repr(exc)}
        elif rank == 0:
            out["cpu_baseline"] = None
        if extra:
            try:
                out["featurize"] = featurize_legs(eng, c)
            except Exception as exc:
                out["featurize"] = {"error": repr(exc)}
            try:
                out["operator_api"] = operator_api_leg(X, c)
            except Exception as exc:
                out["operator_api"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    if multi:
        barrier()
        if comm_kind == "torch":
            dist.destroy_process_group()
        else:
            comm.close()


if __name__ == "__main__":
    main()
