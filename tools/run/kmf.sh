#!/bin/bash
# GPU box: parity of the bf16 filter path, then its timings for several wave staggers
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kmeans_filter.py tests/test_gpu_kmeans.py tests/test_gpu_kmeans_fit.py -x -q > gpurun_out/kmf_t.log 2>&1; rc=$?
tail -15 gpurun_out/kmf_t.log
[ $rc -ne 0 ] && exit $rc
: > gpurun_out/kmf_time.log
for sg in ${STAGGERS:-0 2 4 8 12}; do
  echo "== stagger $sg" >> gpurun_out/kmf_time.log
  MSM_KMEANS_STAGGER=$sg timeout -k 10 300 python tools/time_kmeans_filter.py >> gpurun_out/kmf_time.log 2>&1 || exit 1
done
cat gpurun_out/kmf_time.log
exit 0
