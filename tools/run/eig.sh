#!/bin/bash
# GPU box: TICA / eigensolver tests, then the phase probe of the tridiagonal solver and of msm_tica_solve
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_tica.py tests/test_gpu_revmle.py -x -q > gpurun_out/t.log 2>&1; rc=$?; tail -5 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 60 tools/probe/_bin/tri_probe > gpurun_out/tri_probe.log 2>&1; cat gpurun_out/tri_probe.log
[ "$1" = "tk" ] && { timeout -k 10 120 python tools/time_kernels.py > gpurun_out/tk.log 2>&1; grep -E "tica_solve|eigh|onesided" gpurun_out/tk.log; }
exit 0
