#!/usr/bin/env python3
"""usage: tools/loopstat.py file.hip kernel_substring -> instruction mix of the hottest loop (the one with most MFMAs)."""
import re, subprocess, sys
from pathlib import Path
src = Path(__file__).resolve().parents[1] / "pmarlo_amd" / "csrc" / sys.argv[1]
pat = sys.argv[2]
subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
                "-S", "--cuda-device-only", str(src), "-o", "/tmp/loopstat.s"] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
text = open("/tmp/loopstat.s").read()
for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if pat not in name:
        continue
    lines = [l.strip() for l in m.group(2).splitlines()]
    # find loops: label .. backward branch
    labels = {l[:-1].split(":")[0]: i for i, l in enumerate(lines) if l.startswith(".LBB")}
    print(name[:110])
    for i, l in enumerate(lines):
        mm = re.match(r"s_cbranch_\w+ (\.LBB\S+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            body = lines[labels[mm.group(1)]:i]
            body = [b for b in body if b and not b.startswith((".", ";"))]
            if len(body) < 20:
                continue
            cnt = {}
            for b in body:
                op = b.split()[0]
                key = ("mfma" if "mfma" in op else "accvgpr" if "accvgpr" in op else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
                       "lds" if op.startswith("ds_") else "salu" if op.startswith("s_") else "valu64" if "f64" in op else "valu")
                cnt[key] = cnt.get(key, 0) + 1
            print("   loop @%d instrs: %d %s" % (labels[mm.group(1)], len(body), cnt))
