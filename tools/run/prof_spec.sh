#!/bin/bash
# GPU box: kernel times of one implied-timescale solve (k = 500 and k = 200)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_spec
MSM_SPEC_DEBUG=1 timeout -k 10 120 python3 $R/tools/time_spectrum.py 200 2>&1 | tail -4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_spec -- python3 $R/tools/time_spectrum.py 500 > $R/gpurun_out/prof_spec.log 2>&1
tail -1 $R/gpurun_out/prof_spec.log
f=$(find $R/gpurun_out/prof_spec -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-150
