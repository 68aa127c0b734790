#!/bin/bash
# GPU box: the stock library and N variants on the SAME box, two alternating rounds: abn.sh STAGE v1 v2 ...
mkdir -p gpurun_out; : > gpurun_out/ab.log
stage=$1; shift
for r in 1 2; do
  for v in - "$@"; do
    lib=$v; [ "$v" != "-" ] && lib=tools/probe/_bin/libmsmhip_$v.so
    timeout -k 10 120 python tools/time_variant.py $lib $stage full >> gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
  done
done
cat gpurun_out/ab.log
