#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_tica.py -x -q > gpurun_out/t.log 2>&1; rc=$?; tail -3 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python tools/time_kernels.py > gpurun_out/tk.log 2>&1; grep -E "tica_solve|eigh" gpurun_out/tk.log
