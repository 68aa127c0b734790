#!/bin/bash
# GPU box: parity of the filter path with the stock library, then timings of the stock library and of every variant
# named on the command line (tools/probe/_bin/libmsmhip_NAME.so)
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_kmeans_filter.py tests/test_gpu_kmeans.py tests/test_gpu_kmeans_fit.py -x -q > gpurun_out/kmf_t.log 2>&1; rc=$?
  tail -15 gpurun_out/kmf_t.log
  [ $rc -ne 0 ] && exit $rc
fi
: > gpurun_out/kmf_time.log
for v in - "$@"; do
  lib=$v; [ "$v" != "-" ] && lib=tools/probe/_bin/libmsmhip_$v.so
  timeout -k 10 300 python tools/time_kmeans_filter.py --lib $lib >> gpurun_out/kmf_time.log 2>&1 || { tail -20 gpurun_out/kmf_time.log; exit 1; }
done
cat gpurun_out/kmf_time.log
exit 0
