#!/bin/bash
# GPU box: A/B of the stock library against one variant on the SAME box, three alternating rounds (boxes differ by 5-8 %)
mkdir -p gpurun_out; : > gpurun_out/ab.log
for r in 1 2 3; do
  for v in - "$2"; do
    lib=$v; [ "$v" != "-" ] && lib=tools/probe/_bin/libmsmhip_$v.so
    timeout -k 10 120 python tools/time_variant.py $lib $1 full >> gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
  done
done
cat gpurun_out/ab.log
