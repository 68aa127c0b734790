#!/bin/bash
# GPU box: per-kernel time of the C5-shaped step (1.25 M x 256 f32 per GPU, k = 2000, raw-space clustering)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_c5
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python3 $R/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/prof_c5_bench.log 2>&1 || { tail -5 $R/gpurun_out/prof_c5_bench.log | cut -c1-300; exit 1; }
tail -1 $R/gpurun_out/prof_c5_bench.log | cut -c1-600
find $R/gpurun_out/prof_c5 -name "*_kernel_stats.csv"
