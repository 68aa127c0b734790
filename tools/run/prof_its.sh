#!/bin/bash
# GPU box: kernel trace of the Bayesian ITS scan at the C4 shape (tools/time_its.py)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_its
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_its -- python3 $GRAFT_REPO_ROOT/tools/time_its.py 200 50 100 > $GRAFT_REPO_ROOT/gpurun_out/prof_its.log 2>&1
tail -4 $GRAFT_REPO_ROOT/gpurun_out/prof_its.log
