#!/bin/bash
# GPU box: per-kernel times of count_transitions at C3 (1 M labels, k = 500, lag 10) over five label statistics
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_counts
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_counts -- python3 $R/tools/probe_counts.py > $R/gpurun_out/prof_counts.log 2>&1 || exit 1
grep "median" $R/gpurun_out/prof_counts.log
f=$(find $R/gpurun_out/prof_counts -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-160
