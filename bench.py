#!/usr/bin/env python3
"""Benchmark of the hot path: featurised shard -> TICA -> k-means -> transition matrix.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched by
torch.distributed.run with one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric "... 1Mx64 synth"): per GPU one synthetic shard of
1,000,000 frames x 64 float32 features (AR(1)-latent generator of the reference's
tests/perf/test_tica_perf.py:65-81, vectorised; seed 1000 + rank), already resident in
HBM when the timed region starts.  One step = one pass of the whole path over the shard:
fp64-MFMA lagged covariance (the standardisation sums come out of it) -> on-device TICA solve
(dim 10) -> projection (with max |Y| for the fixed-point scale) -> k-means (k = 500, seeded init + 10 full-batch Lloyd iterations, fp64-MFMA
assignment) -> final assignment -> lag-10 transition counts -> row-normalised T.
Weak scaling: every rank holds its own shard; only the small moment / count / centre
buffers are all-reduced (pmarlo_amd/dist.py).
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

N_FRAMES, N_FEATURES, TICA_DIM, K_STATES, LAG, KMEANS_ITERS = 1_000_000, 64, 10, 500, 10, 10
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_MFMA_PEAK_TF = 78.6       # MI355X fp64 matrix (= vector) peak
# HBM-side bytes per launch of the dominant kernel at the default workload, from the PMC passes in
# profiles/r01_pmc_hbm.md (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of the guide):
# reads of Y (80 MB) + the atomics flush of the member sums
KMEANS_ACCUM_TRAFFIC_BYTES = (2 * 39293 + 10667) * 1024


def cpu_baseline(X: np.ndarray) -> dict:
    """The oracle's numpy / scikit-learn restatement of the same path (what the reference runs at
    this size: _preprocess -> TICA -> MiniBatchKMeans branch (n*d >= 5e6) -> predict ->
    _weighted_counts -> _normalise_counts), timed once on this host."""
    from oracle import npport

    t0 = time.perf_counter()
    Xp = npport.preprocess(X, scale=True)
    model = npport.tica_fit([Xp], LAG, dim=TICA_DIM)
    Y = npport.tica_transform(model, Xp)
    fit = npport.kmeans_discretizer_fit(Y, K_STATES, random_state=0)
    labels = npport.kmeans_predict((Y - fit["mean"]) / fit["std_safe"], fit["centers"])
    counts, _ = npport.weighted_counts(labels, K_STATES, LAG)
    npport.normalise_counts(counts)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": X.shape[0] / dt, "unit": "frames/s", "cores": int(threads), "kind": "port",
            "sample": f"the full {X.shape[0]}x{X.shape[1]} shard once, {dt:.1f} s "
                      "(numpy TICA + sklearn MiniBatchKMeans + predict + numpy counts)",
            "tica_eigenvalues": model["eigenvalues"][:TICA_DIM].tolist()}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-single-thread", action="store_true",
                    help="also time the CPU restatement pinned to one thread on the full shard (about a minute)")
    ap.add_argument("--frames", type=int, default=N_FRAMES, help="frames per GPU (default: the BASELINE config)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_FORCE_EXCHANGE=1 (under torch.distributed.run with one rank): keep every collective in the
    # step, so a one-GPU box exercises the RCCL calls the N > 1 runs make
    multi = world > 1 or os.environ.get("BENCH_FORCE_EXCHANGE") == "1"
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from pmarlo_amd.device import Engine
    from pmarlo_amd.dist import ShardConfig, ShardedMSM, TorchComm, torch_exchange_buffers
    from tests import _gen

    comm = None
    shared = None
    if multi:
        import torch
        import torch.distributed as dist

        # rehearsal knobs (a one-GPU box cannot host two RCCL ranks): BENCH_BACKEND=gloo and
        # BENCH_SAME_GPU=1 run every rank on device 0 with gloo moving the (device) buffers
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if os.environ.get("BENCH_SAME_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)
        # one explicit stream shared by the engine's kernels and torch's collectives (the default
        # stream's handle is 0, which the C ABI reads as "create your own stream")
        stream = torch.cuda.Stream(device=local_rank)
        torch.cuda.set_stream(stream)
        eng = Engine(local_rank, stream=stream.cuda_stream)
    else:
        eng = Engine(local_rank)

    n = int(args.frames)
    cfg = ShardConfig(n_frames=n, n_features=N_FEATURES, tica_dim=TICA_DIM, k=K_STATES, lag=LAG,
                      kmeans_iters=KMEANS_ITERS, seed=0, n_total=n * world)
    X = _gen.correlated_series(n, N_FEATURES, seed=1000 + rank)
    xd = eng.to_device(X)
    if multi:
        tensors, shared = torch_exchange_buffers(eng, cfg, torch.device("cuda", local_rank))
        comm = TorchComm(tensors)
    msm = ShardedMSM(eng, cfg, xd, comm=comm, shared=shared, always_exchange=multi)

    def barrier():
        eng.sync()
        if multi:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(0, args.warmup)):
        msm.step()
    barrier()
    msm.time_accum = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        msm.step()
    barrier()
    elapsed = time.perf_counter() - t0
    msm.time_accum = False
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # from-host pass: the same step with the shard uploaded from pageable host
    # memory inside the clock -- reported beside `value`, never as it
    from_host = None
    if world == 1 and not multi:
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
        from_host = {"value": n / tot, "unit": "frames/s", "h2d_ms": h2d * 1e3, "step_ms": (tot - h2d) * 1e3,
                     "h2d_GBps": X.nbytes / h2d / 1e9, "note": "one pageable-memory upload + one step"}

    # dominant kernel: k-means assign+accumulate (fp64 MFMA), measured with HIP events on the
    # engine's stream inside the timed region
    acc_ms = [a.elapsed_ms(b) for a, b in msm.accum_events]
    acc_ms_avg = float(np.mean(acc_ms)) if acc_ms else float("nan")
    flops_per_launch = 2.0 * K_STATES * TICA_DIM * n            # 2*k*d per frame
    achieved_tf = flops_per_launch / (acc_ms_avg * 1e-3) / 1e12

    if rank == 0:
        total_frames = float(n) * world * args.steps
        out = {
            "metric": "frames/sec featurize->TICA->k-means->T-matrix, 1Mx64 synth; ITS rel-err",
            "value": total_frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"synthetic {n} frames x {N_FEATURES} f32 features per GPU (AR(1) latents), "
                                   f"TICA->{TICA_DIM}, k={K_STATES} ({KMEANS_ITERS} Lloyd iterations), lag={LAG}, "
                                   "row-normalised T", "frames_per_gpu": n, "features": N_FEATURES,
                       "tica_dim": TICA_DIM, "k": K_STATES, "lag": LAG, "kmeans_iters": KMEANS_ITERS,
                       "parallelism": f"shards{world}",
                       "exchange": ("rccl" if os.environ.get("BENCH_BACKEND", "nccl") == "nccl" else "gloo") if multi
                       else None},
            "roofline": {"kernel": "kmeans_mfma_kernel<double,3,2,1024,true,true,false> (assign + accumulate)", "bound": "mfma",
                         "achieved": achieved_tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_MFMA_PEAK_TF,
                         "traffic": KMEANS_ACCUM_TRAFFIC_BYTES if n == N_FRAMES else None,
                         "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/r01_pmc_hbm.md)",
                         "launch_ms": acc_ms_avg, "launches_timed": len(acc_ms),
                         "algorithmic_flops_per_launch": flops_per_launch},
        }
        # ---- parity of this very run against the CPU oracle (outside the timed region) ----
        parity = {}
        try:
            from oracle import cport, npport

            counts = msm.buf["counts"].view((K_STATES, K_STATES)).to_host()
            pairs = int(msm.buf["counts"].view((1,), offset_elems=K_STATES * K_STATES).to_host()[0])
            labels = msm.labels.to_host()
            if world == 1:
                want, pw = cport.count_transitions(labels, K_STATES, LAG)
                parity["counts_bit_exact"] = bool(np.array_equal(counts, want) and pairs == pw)
                want_lab = cport.kmeans_assign(msm.Y.to_host(), msm.buf["centers"].to_host())
                parity["labels_bit_exact_given_centres"] = bool(np.array_equal(labels, want_lab))
            tm = eng.transition_matrix(eng.to_device(counts), mode=1)
            spec = eng.spectrum(tm["T"], n=tm["n_active"], n_its=5, lags=[float(LAG)], allow_unconverged=True)
            ev_ref, ts_ref = npport.its_from_counts(counts, LAG, 5)
            ok = np.isfinite(ts_ref)
            parity["its_rel_err"] = float(np.max(np.abs(spec["its_ts"][0][ok] - ts_ref[ok]) / ts_ref[ok]))
            parity["its_timescales_frames"] = spec["its_ts"][0].tolist()
            parity["its_residual"] = float(spec["residual"][0])
            parity["tica_eigenvalues"] = msm.eig.to_host()[:TICA_DIM].tolist()
            parity["tica_rank"] = int(msm.rank_d.to_host()[0])
            if multi and n * world <= 2_000_000:
                # rank 0 regenerates every shard (seeded) and checks the all-reduced TICA against the
                # oracle on the list of shards (pairs never cross a shard)
                shards = [npport.preprocess(np.vstack([_gen.correlated_series(n, N_FEATURES, seed=1000 + r)
                                                       for r in range(world)]), scale=True)]
                parts = [shards[0][r * n:(r + 1) * n] for r in range(world)]
                ref = npport.tica_fit(parts, LAG, dim=TICA_DIM)
                got = np.asarray(parity["tica_eigenvalues"])
                m = min(2 * world, TICA_DIM)  # the resolved slow modes (2 per independently mixed shard)
                parity["tica_eig_rel_err"] = float(np.max(np.abs(got[:m] - ref["eigenvalues"][:m])
                                                          / np.abs(ref["eigenvalues"][:m])))
                parity["tica_rank_oracle"] = int(ref["rank"])
        except Exception as exc:  # parity reporting must not hide the throughput line
            parity["error"] = repr(exc)
        out["parity"] = parity
        out["from_host"] = from_host
        if not multi and not args.no_cpu_baseline:
            cb = cpu_baseline(X)
            ref_eig = np.asarray(cb.pop("tica_eigenvalues"))
            got = np.asarray(parity.get("tica_eigenvalues", ref_eig))
            out["parity"]["tica_eig_rel_err"] = float(np.max(np.abs(got - ref_eig) / np.abs(ref_eig)))
            out["cpu_baseline"] = cb
            if args.cpu_single_thread:
                from threadpoolctl import threadpool_limits

                with threadpool_limits(limits=1):
                    cb1 = cpu_baseline(X)
                cb1.pop("tica_eigenvalues", None)
                cb1["cores"] = 1
                out["cpu_baseline_1thread"] = cb1
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
