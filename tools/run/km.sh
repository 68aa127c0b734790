#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_kmeans_fit.py -x -q > gpurun_out/t.log 2>&1; rc=$?; tail -8 gpurun_out/t.log
exit $rc
