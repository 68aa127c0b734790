#!/bin/bash
# GPU box: WRITE_SIZE / FETCH_SIZE of the k-means assign pass for the stock library and the variants named
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in - "$@"; do
  lib=$v; [ "$v" != "-" ] && lib=tools/probe/_bin/libmsmhip_$v.so
  for c in WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmcv_${v}_$c
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcv_${v}_$c -- python3 $R/tools/time_variant.py $lib kmeans > $R/gpurun_out/pmcv_${v}_$c.log 2>&1 || { tail -3 $R/gpurun_out/pmcv_${v}_$c.log; exit 1; }
    python3 - "$R/gpurun_out/pmcv_${v}_$c" "$v" "$c" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    by[r['Kernel_Name'].split('(')[0][:70]].append(float(r['Counter_Value']))
for k, v in by.items():
    if 'filter' in k or 'pack' in k: print(sys.argv[2], sys.argv[3], k, len(v), 'mean KiB', round(sum(v) / len(v)))
PY
  done
done
