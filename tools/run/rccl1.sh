#!/bin/bash
# GPU box: the full-size step with every collective issued through RCCL (one rank), next to the plain step
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_plain.log 2>&1 || { tail -20 gpurun_out/b_plain.log; exit 1; }
BENCH_FORCE_EXCHANGE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_rccl1.log 2>&1 || { tail -20 gpurun_out/b_rccl1.log; exit 1; }
python - <<'PY'
import json
for f in ("b_plain", "b_rccl1"):
    ln = [l for l in open(f"gpurun_out/{f}.log") if l.startswith("{")][-1]
    o = json.loads(ln)
    print(f, round(o["ms_per_step"], 3), "ms/step", round(o["value"] / 1e6, 1), "M frames/s", "accum us", round(o["roofline"]["launch_ms"] * 1e3, 1))
PY
