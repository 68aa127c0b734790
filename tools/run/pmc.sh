#!/bin/bash
# GPU box: HBM-side byte counters of the bench kernels.  FETCH_SIZE (3 TCC slots) and WRITE_SIZE
# (2 slots) do not fit one pass: two runs, --pmc with --kernel-trace only.
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log | cut -c1-300; exit 1; }
  echo "$c done"
done
find $GRAFT_REPO_ROOT/gpurun_out/pmc_* -name "*.csv" | head
