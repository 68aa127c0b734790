#!/bin/bash
# GPU box: time STAGE ($1) with the stock library and every variant named after it
mkdir -p gpurun_out
stage=$1; shift
: > gpurun_out/variants.log
for v in - "$@"; do
  lib=$v; [ "$v" != "-" ] && lib=tools/probe/_bin/libmsmhip_$v.so
  timeout -k 10 120 python tools/time_variant.py $lib $stage >> gpurun_out/variants.log 2>&1 || { tail -5 gpurun_out/variants.log; exit 1; }
done
cat gpurun_out/variants.log
