#!/usr/bin/env python3
"""usage: tools/kres.py file.hip -> per-kernel register / spill / occupancy table (gfx950)."""
import re
import subprocess
import sys
from pathlib import Path

src = Path(__file__).resolve().parents[1] / "pmarlo_amd" / "csrc" / sys.argv[1]
cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
       "-c", str(src), "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (?:[^ ]+:\d+:\d+: )?\s*(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()[:100]
    print("%-100s V=%s A=%s S=%s spill=%s scratch=%s occ=%s lds=%s" % (
        name, r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("VGPRs Spill"),
        r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
