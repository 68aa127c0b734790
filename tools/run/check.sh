#!/bin/bash
# GPU box: full -m gpu suite, then a bench line and per-kernel timings (outputs under gpurun_out/)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; rc=$?; tail -5 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b.log 2>&1 || { tail -20 gpurun_out/b.log; exit 1; }
tail -1 gpurun_out/b.log
timeout -k 10 120 python tools/time_kernels.py > gpurun_out/tk.log 2>&1; tail -30 gpurun_out/tk.log
