#!/bin/bash
# GPU box: every k-means / sharded-step test, then the k-means timings and a bench line
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kmeans_filter.py tests/test_gpu_kmeans.py tests/test_gpu_kmeans_fit.py tests/test_gpu_configs.py tests/test_gpu_rccl.py tests/test_gpu_api.py -x -q > gpurun_out/t.log 2>&1; rc=$?; tail -8 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/time_kmeans_filter.py > gpurun_out/kmf_time.log 2>&1; tail -7 gpurun_out/kmf_time.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/b.log 2>&1 || { tail -20 gpurun_out/b.log; exit 1; }
tail -1 gpurun_out/b.log | cut -c1-400; tail -1 gpurun_out/b.log | grep -o '"stages_ms": {[^}]*}'
