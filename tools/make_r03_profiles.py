#!/usr/bin/env python3
"""gpurun_out/prof_r03 + pmc_r03_* (tools/run/prof_r03.sh) -> profiles/r03_bench_kernel_stats.{csv,md}, r03_pmc.md,
r03_pmc_hbm.{md,json}."""
import glob
import json
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
P = ROOT / "profiles"
CMD = "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs"


def main():
    stats = sorted(glob.glob(str(ROOT / "gpurun_out/prof_r03/*/*_kernel_stats.csv")), key=lambda f: -Path(f).stat().st_mtime)[0]
    shutil.copy(stats, P / "r03_bench_kernel_stats.csv")
    subprocess.run([sys.executable, str(ROOT / "tools/prof_summary.py"), str(P / "r03_bench_kernel_stats.csv"),
                    str(P / "r03_bench_kernel_stats.md"), "Bench step, round 3 (one MI355X), end of round", CMD,
                    "7 steps (2 warm-up + 5 timed) + the parity block's calls; tools/run/prof_r03.sh.  New this round: the scratch-free "
                    "12-wave k-means filter with the compact 80-byte frame image and the in-LDS fp32 scan, the two-pass transition "
                    "count without global atomics, the persistent subspace iteration of the implied-timescale solve."], check=True, capture_output=True)
    dirs = [str(ROOT / f"gpurun_out/pmc_r03_{i}") for i in range(4)]
    table = subprocess.run([sys.executable, str(ROOT / "tools/pmc_summary.py"), *dirs, "--json", "/tmp/pmc.json"],
                           check=True, capture_output=True, text=True).stdout
    d = json.load(open("/tmp/pmc.json"))

    def util(k):
        v = d[k]
        return v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)

    names = {k.split("<")[0]: k for k in d}
    u = {n: util(names[n]) for n in ("kmeans_filter_kernel", "cov_fused_kernel", "project_mfma_kernel") if n in names}
    head = ["# Round 3: PMC counters of the bench step (per launch means), end of round", "",
            "Command per counter set (`tools/run/prof_r03.sh`; counters only ever with `--kernel-trace`): `rocprofv3 --pmc <set> "
            "--kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs`.",
            "Sets: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU "
            "SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY "
            "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_* SQ_VALU_MFMA_COEXEC_CYCLES.",
            "FETCH_SIZE / WRITE_SIZE in KiB (reads = 2 x FETCH_SIZE on gfx950, see r03_pmc_hbm.md).  Matrix-pipe utilisation of a "
            "kernel = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): the busy cycles are summed over the "
            "SIMDs, the GUI-active cycles over the XCDs (" + ", ".join(f"{n} {v:.2f}" for n, v in u.items()) + ").  "
            "SQ_VALU_MFMA_COEXEC_CYCLES is zero for the fp64 matrix kernels: fp64 vector and fp64 matrix instructions do not "
            "overlap on this chip.", ""]
    (P / "r03_pmc.md").write_text("\n".join(head) + table)
    alg = {"kmeans_filter_kernel<double, 2, 4, true, false>": 168, "kmeans_filter_kernel<double, 2, 4, false, false>": 164,
           "cov_fused_kernel<float, 4, true, true, true, true>": 256, "cov_fused_kernel<float, 4, true, true, true>": 256, "project_mfma_kernel<float, true>": 336, "project_mfma_kernel<float, true, true>": 336,
           "kmeans_pack_kernel<double, 10>": 160, "count_lds_kernel<false>": 6, "count_bucket_scatter_kernel": 6, "count_bucket_bin_kernel": 4, "moments_partial_kernel<float>": 256}
    rows = [(k, v["dispatches"], v["FETCH_SIZE"], v["WRITE_SIZE"], (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 1e6)
            for k, v in d.items() if v.get("FETCH_SIZE") is not None and v.get("WRITE_SIZE") is not None]
    out = ["# Round 3: HBM-side traffic per launch (PMC), end of round", "",
           "Command (two passes, FETCH_SIZE needs 3 TCC slots and WRITE_SIZE 2): `rocprofv3 --pmc FETCH_SIZE --kernel-trace "
           "--output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs`, then the same with "
           "`--pmc WRITE_SIZE` (`tools/run/prof_r03.sh`).", "",
           "Counters are in KiB.  Correction per `MI355X_MICROARCH.md` (HBM section), calibrated in round 1 (`r01_pmc_hbm.md`): on "
           "gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so reads = 2 x FETCH_SIZE; WRITE_SIZE is taken as is.  "
           "Infinity-Cache hits are counted, so this is fabric traffic, an upper bound on HBM bytes.", "",
           "| kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic = 2 F + W (MB) | algorithmic (MB) |",
           "|---|---:|---:|---:|---:|---:|"]
    for k, n, f, w, mb in sorted(rows, key=lambda r: -r[4]):
        if mb >= 1 or "count_" in k:
            out.append(f"| `{k}` | {n} | {f:.0f} | {w:.0f} | {mb:.1f} | {alg.get(k, alg.get(k.split('(')[0].split('::')[-1], ''))} |")
    out += ["", "Algorithmic bytes: k-means passes read the frame images (80 MB), Y (80 MB: the exact candidate pick) and the old label, "
            "and write label (+ distance); the image build reads Y and writes the images; the count passes read the labels "
            "(4 MB), write and re-read the 16-bit elements (2 MB + offsets) and write the int64 matrix (2 MB); the covariance pass reads X once (256 MB); the projection reads X and writes "
            "Y (80 MB).", ""]
    (P / "r03_pmc_hbm.md").write_text("\n".join(out))
    dom = "kmeans_filter_kernel<double, 2, 4, true, false>"
    v = d[dom]
    json.dump({"kernel_prefix": "kmeans_filter_kernel<double,2,4,true,false>", "kernel": dom, "fetch_size_kib": v["FETCH_SIZE"],
               "write_size_kib": v["WRITE_SIZE"], "bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
               "algorithmic_bytes_per_launch": 168e6,
               "source": "profiles/r03_pmc_hbm.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two passes; reads = 2 x FETCH_SIZE on gfx950)"},
              open(P / "r03_pmc_hbm.json", "w"), indent=1)
    print((P / "r03_bench_kernel_stats.md").read_text()[:2400])
    print("\n".join(out[6:14]))
    print(u)


if __name__ == "__main__":
    main()
