#!/bin/bash
# GPU box: the RCCL / two-rank tests, then the default bench line (all legs)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_rccl.py -x -q > gpurun_out/rccl_t.log 2>&1; rc=$?
tail -30 gpurun_out/rccl_t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python bench.py > gpurun_out/bench_default.log 2>&1; rc=$?
tail -c 6000 gpurun_out/bench_default.log
exit $rc
