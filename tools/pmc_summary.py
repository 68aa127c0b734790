#!/usr/bin/env python3
"""usage: tools/pmc_summary.py <dir with pmc_*/.../*_counter_collection.csv ...> -> per-kernel mean of every counter
per dispatch (markdown on stdout, JSON with --json FILE)."""
import csv
import glob
import json
import subprocess
import sys
from collections import defaultdict


def demangle(n):
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    return d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    files = []
    for a in args:
        files += glob.glob(a + "/**/*_counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in files:
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    names = {k: demangle(k) for k in acc}
    counters = sorted({c for k in acc for c in acc[k]})
    out = {}
    rows = sorted(acc, key=lambda k: -sum(dur[k]))
    print("| kernel | dispatches | avg us (profiled) | " + " | ".join(counters) + " |")
    print("|---|---:|---:|" + "---:|" * len(counters))
    for k in rows:
        if sum(dur[k]) < 50:
            continue
        vals = {c: (sum(acc[k][c]) / len(acc[k][c]) if acc[k][c] else None) for c in counters}
        out[names[k]] = {"dispatches": len(dur[k]), "avg_us": sum(dur[k]) / len(dur[k]), **vals}
        print(f"| `{names[k]}` | {len(dur[k])} | {sum(dur[k]) / len(dur[k]):.1f} | "
              + " | ".join("" if vals[c] is None else f"{vals[c]:.4g}" for c in counters) + " |")
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
