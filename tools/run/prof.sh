#!/bin/bash
# GPU box: rocprofv3 kernel trace of the bench (outputs under gpurun_out/prof)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log | cut -c1-300
find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*kernel_stats.csv" | head
