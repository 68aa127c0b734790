#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 180 python -m pytest tests/test_gpu_tica.py tests/test_gpu_configs.py -x -q > gpurun_out/cov_t.log 2>&1; rc=$?
tail -5 gpurun_out/cov_t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python tools/time_kernels.py 2>&1 | grep -E "lagged_moments|project|tica_solve" || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_cov
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_cov -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $GRAFT_REPO_ROOT/gpurun_out/pmc_cov.log 2>&1 || { tail -3 $GRAFT_REPO_ROOT/gpurun_out/pmc_cov.log; exit 1; }
cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py gpurun_out/pmc_cov | grep -E "cov_fused|kernel \|" | cut -c1-200
