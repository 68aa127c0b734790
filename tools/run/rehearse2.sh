#!/bin/bash
# two ranks on the one GPU, gloo moving the exchange buffers: rehearsal of the N > 1 bench path
mkdir -p gpurun_out
export BENCH_COMM=torch BENCH_BACKEND=gloo BENCH_SAME_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --frames 500000 > gpurun_out/rehearse2.log 2>&1
rc=$?; tail -2 gpurun_out/rehearse2.log | cut -c1-1500; exit $rc
