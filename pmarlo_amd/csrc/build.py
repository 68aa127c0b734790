"""Build libmsmhip.so (gfx950) in-tree with hipcc.

Usage: ``python -m pmarlo_amd.csrc.build [--force]``.  hipcc cross-compiles
without a GPU; the resulting .so is git-ignored but travels with the tree.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB = HERE / "libmsmhip.so"
OBJ_DIR = HERE / "build"
ARCH = "gfx950"
FLAGS = [
    "-O3",
    "-std=c++17",
    f"--offload-arch={ARCH}",
    "-fPIC",
    "-ffp-contract=off",       # fma() is written where it is meant; nothing else fuses
    "-munsafe-fp-atomics",     # hardware fp64 atomic add (LDS + global)
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build libmsmhip.so")
    return exe


def _sources() -> list[Path]:
    return sorted(HERE.glob("*.hip"))


def _deps_mtime() -> float:
    hdrs = list(HERE.glob("*.h")) + [HERE.parent.parent / "include" / "msmhip.h"]
    return max(p.stat().st_mtime for p in hdrs)


def _compile(src: Path, force: bool) -> Path:
    obj = OBJ_DIR / (src.stem + ".o")
    newest = max(src.stat().st_mtime, _deps_mtime())
    if not force and obj.exists() and obj.stat().st_mtime >= newest:
        return obj
    cmd = [_hipcc(), *FLAGS, "-c", str(src), "-o", str(obj)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
    if res.stderr.strip():
        sys.stderr.write(res.stderr)
    return obj


def build(force: bool = False, jobs: int | None = None) -> Path:
    OBJ_DIR.mkdir(exist_ok=True)
    srcs = _sources()
    if not srcs:
        raise RuntimeError(f"no .hip sources in {HERE}")
    jobs = jobs or min(4, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(lambda s: _compile(s, force), srcs))
    newest_obj = max(o.stat().st_mtime for o in objs)
    if force or not LIB.exists() or LIB.stat().st_mtime < newest_obj:
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv)
    print(path)
