"""The kernels of the bench step must not touch scratch memory.

A per-lane select written as `q == u ? a[u] : ...` once became an indexed access of arrays in SCRATCH (64 B of extra HBM
writes per frame, found only through the WRITE_SIZE counter); register spills in a hot loop would hide the same way.
The check reads the `.private_segment_fixed_size` the compiler recorded for every kernel of the built library."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

from pmarlo_amd._lib import LIB_PATH

LLVM = Path("/opt/rocm/lib/llvm/bin")

# demangled-name patterns of the bench step's kernels (and their siblings of the same template)
HOT = [r"kmeans_filter_kernel<", r"kmeans_pack_kernel<", r"kmeans_update_kernel", r"tica_solve_kernel<", r"eigh_kernel",
       r"project_mfma_kernel<", r"count_lds_kernel<", r"cov_reduce_kernel<", r"moments_from_lagged_kernel<",
       r"cov_fused_kernel<float, 4, true, true, (true|false), true>"]   # the symmetric flavour on the vector path


def _kernel_scratch(tmp_path):
    lib = tmp_path / "lib.so"
    shutil.copy(LIB_PATH, lib)
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(lib)], cwd=tmp_path, check=True, capture_output=True)
    out = {}
    for co in sorted(tmp_path.glob("lib.so.*gfx950*")):
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s*\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name:
                out[name] = int(m.group(1))
                name = None
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return {d: out[n] for n, d in zip(names, dem)}


@pytest.mark.skipif(not (LLVM / "llvm-readelf").exists() or shutil.which("c++filt") is None or not LIB_PATH.exists(),
                    reason="needs the ROCm llvm tools and the built library")
def test_bench_step_kernels_use_no_scratch(tmp_path):
    scratch = _kernel_scratch(tmp_path)
    assert len(scratch) > 100                       # the whole library was read
    seen = {p: 0 for p in HOT}
    offenders = []
    for name, size in scratch.items():
        for p in HOT:
            if re.search(p, name):
                seen[p] += 1
                if size:
                    offenders.append((name, size))
    assert all(seen.values()), f"patterns that matched no kernel: {[p for p, c in seen.items() if not c]}"
    assert not offenders, f"scratch memory in hot kernels: {offenders}"
