#!/bin/bash
# GPU box: kernel times of the C4 lag scan with posterior samples (tools/time_its.py)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_its
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_its -- python3 $R/tools/time_its.py > $R/gpurun_out/prof_its.log 2>&1
tail -4 $R/gpurun_out/prof_its.log | cut -c1-200
f=$(find $R/gpurun_out/prof_its -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-200
exit 0
