#!/bin/bash
# GPU box: kernel times of the two count passes at C3 for several pass-1 chunk sizes
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 4096 2048 1024; do
  rm -rf $R/gpurun_out/prof_counts_$c
  MSM_COUNTS_CHUNK=$c timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_counts_$c -- python3 $R/tools/probe_counts.py > $R/gpurun_out/prof_counts_$c.log 2>&1 || exit 1
  echo "chunk $c"; grep median $R/gpurun_out/prof_counts_$c.log | head -2
  f=$(find $R/gpurun_out/prof_counts_$c -name "*kernel_stats.csv" | head -1); head -3 $f | cut -d, -f1-4 | cut -c1-60,160-220
done
