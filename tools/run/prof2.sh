#!/bin/bash
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log | cut -c1-200
