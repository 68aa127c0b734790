#!/usr/bin/env python3
"""gpurun_out/prof_spec, prof_its, prof_counts (tools/run/prof_spec.sh, prof_its.sh, prof_counts.sh) ->
profiles/r03_its_spectrum_kernel_stats.{csv,md}, r03_its_scan_kernel_stats.{csv,md}, r03_counts_kernel_stats.{csv,md}."""
import glob
import re
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
P = ROOT / "profiles"


def newest(pattern):
    return sorted(glob.glob(str(ROOT / pattern)), key=lambda f: -Path(f).stat().st_mtime)[0]


def result_lines(log, pat):
    return [ln.strip() for ln in Path(log).read_text().splitlines() if re.search(pat, ln)]


def emit(src_dir, name, title, cmd, note):
    shutil.copy(newest(f"gpurun_out/{src_dir}/*/*_kernel_stats.csv"), P / f"{name}.csv")
    subprocess.run([sys.executable, str(ROOT / "tools/prof_summary.py"), str(P / f"{name}.csv"), str(P / f"{name}.md"), title, cmd,
                    note], check=True, capture_output=True)
    print((P / f"{name}.md").read_text()[:1500])


def main():
    spec = result_lines(ROOT / "gpurun_out/prof_spec.log", r"ms wall per solve")
    emit("prof_spec", "r03_its_spectrum_kernel_stats", "Round 3: one implied-timescale solve, k = 500 microstates, 5 timescales",
         "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/time_spectrum.py 500 (tools/run/prof_spec.sh)",
         "Three solves (the script repeats the call) of a metastable 6-block chain; each solve = 2 squarings (T^4), the seeding "
         "step, ONE launch of the persistent subspace iteration on T^4 (6 iterations) and one finishing launch "
         "(product with T + Rayleigh-Ritz, residuals of real and complex pairs).  " + " ".join(spec))
    its = result_lines(ROOT / "gpurun_out/prof_its.log", r"k=200 L=50|numpy eigvals")
    emit("prof_its", "r03_its_scan_kernel_stats", "Round 3: Bayesian ITS scan (k = 200, 50 lags x 100 posterior samples = 5000 matrices)",
         "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/time_its.py (tools/run/prof_its.sh)",
         "Includes the deterministic scan (50 matrices) and the warm-ups of tools/time_its.py.  " + " | ".join(its))
    cnt = result_lines(ROOT / "gpurun_out/prof_counts.log", r"median")
    emit("prof_counts", "r03_counts_kernel_stats", "Round 3: transition counts at C3 (1 M labels, k = 500, lag 10), five label statistics",
         "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/probe_counts.py (tools/run/prof_counts.sh)",
         "12 calls per label statistic (random, constant, runs of 16, 4 states, slow walk); HIP-event times of the whole call "
         "(both launches): " + " | ".join(cnt))


if __name__ == "__main__":
    main()
