#!/usr/bin/env python
"""Run bench.py with library debug knobs set first (A/B experiments), e.g.
    python tools/bench_knobs.py few_rows_max=8 gemv_rows=8 -- --batch 8 --steps 3 --no-cpu-baseline --no-roofline"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (libbt_hip must load after torch)
from block_transformer_amd import _lib  # noqa: E402
from block_transformer_amd._lib import lib  # noqa: E402

_lib.use_debug_library(True)   # the bt_debug_* knobs exist only in libbt_hip_dbg.so

sep = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
for kv in sys.argv[1:sep]:
    k, v = kv.split("=")
    getattr(lib(), "bt_debug_" + k)(int(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[sep + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
