#!/usr/bin/env python3
"""bench.py with a variant build of the library (tools/build_variant.sh): tools/bench_variant.py LIB [bench arguments]."""
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import pmarlo_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = (ROOT / sys.argv[1]).resolve()
sys.argv = [str(ROOT / "bench.py"), *sys.argv[2:]]
runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
