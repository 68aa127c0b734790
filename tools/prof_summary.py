#!/usr/bin/env python3
"""usage: tools/prof_summary.py <kernel_stats.csv> <out.md> "<title>" "<command>" ["note"] -> markdown table"""
import csv, subprocess, sys
src, out, title, cmd = sys.argv[1:5]
note = sys.argv[5] if len(sys.argv) > 5 else ""
rows = list(csv.DictReader(open(src)))
with open(out, "w") as f:
    f.write(f"# {title}\n\nCommand: `{cmd}`\n\n{note}\n\n| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows:
        dem = subprocess.run(["c++filt", r["Name"]], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").replace("void ", "")
        dem = dem.split("(")[0][:80]
        f.write(f"| `{dem}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
print(open(out).read())
