#!/bin/bash
# GPU box: tests named on the command line (default: TICA + API), then per-kernel timings
mkdir -p gpurun_out
T="${@:-tests/test_gpu_tica.py tests/test_gpu_api.py}"
timeout -k 10 600 python -m pytest $T -x -q > gpurun_out/t.log 2>&1; rc=$?; tail -3 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python tools/time_kernels.py > gpurun_out/tk.log 2>&1; tail -14 gpurun_out/tk.log
