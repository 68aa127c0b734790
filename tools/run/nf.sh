#!/bin/bash
# GPU box: k-means parity tests and a bench line (used when trying kernel variants)
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_kmeans_fit.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_nf.log 2>&1 || { tail -5 gpurun_out/b_nf.log; exit 1; }
python - <<'PY'
import json
o = json.loads([l for l in open("gpurun_out/b_nf.log") if l.startswith("{")][-1])
print(o["ms_per_step"], o["value"] / 1e6, o["roofline"]["launch_ms"], o["roofline"]["frac"], o["parity"]["counts_bit_exact"],
      o["parity"]["labels_bit_exact_given_centres"])
PY
