#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_api.py -x -q > gpurun_out/t.log 2>&1; rc=$?; tail -5 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b.log 2>&1 || { tail -20 gpurun_out/b.log; exit 1; }
tail -1 gpurun_out/b.log | cut -c1-200
