#!/bin/bash
mkdir -p gpurun_out
export BENCH_COMM=torch BENCH_BACKEND=gloo BENCH_SAME_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 4 --steps 2 --warmup 1 --frames 250000 > gpurun_out/rehearse4.log 2>&1
rc=$?; tail -1 gpurun_out/rehearse4.log | cut -c1-1200; exit $rc
