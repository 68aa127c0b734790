#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 python tools/time_variant.py - cov > gpurun_out/covsym.log 2>&1 || { tail -5 gpurun_out/covsym.log; exit 1; }
timeout -k 10 120 python tools/time_variant.py - covsym >> gpurun_out/covsym.log 2>&1 || { tail -5 gpurun_out/covsym.log; exit 1; }
cat gpurun_out/covsym.log
timeout -k 10 600 python -m pytest tests/test_gpu_tica.py -x -q > gpurun_out/t.log 2>&1; rc=$?; tail -15 gpurun_out/t.log; exit $rc
