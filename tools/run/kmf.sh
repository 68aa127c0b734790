#!/bin/bash
# GPU box: parity of the bf16 filter path, then its timings
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kmeans_filter.py tests/test_gpu_kmeans.py tests/test_gpu_kmeans_fit.py -x -q > gpurun_out/kmf_t.log 2>&1; rc=$?
tail -15 gpurun_out/kmf_t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/time_kmeans_filter.py > gpurun_out/kmf_time.log 2>&1 || { tail -20 gpurun_out/kmf_time.log; exit 1; }
cat gpurun_out/kmf_time.log
exit 0
