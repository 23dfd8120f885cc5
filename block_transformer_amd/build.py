"""Build libbt_hip.so (gfx950) from block_transformer_amd/csrc/*.hip with hipcc — in-tree, so the .so travels with the repo.

    python -m block_transformer_amd.build [--force] [--no-debug]

Two libraries from the same sources:
  libbt_hip.so       the product: no bt_debug_* hooks, no A/B or timing-ablation kernels, no mutable dispatch state;
  libbt_hip_dbg.so   -DBT_DEBUG_HOOKS: the same kernels plus the hooks — loaded ONLY by tests/ and tools/kernel_probe.py
                     (block_transformer_amd._lib.use_debug_library()).

hipcc cross-compiles for gfx950 without a GPU.  Objects are cached per source under csrc/_build and rebuilt when the
source (or a header) is newer.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libbt_hip.so")
LIB_DBG = os.path.join(HERE, "libbt_hip_dbg.so")
SOURCES = ["core.hip", "gemm.hip", "norm.hip", "misc.hip", "attention.hip", "engine.hip"]
HEADERS = [os.path.join(CSRC, "bt_common.h"), os.path.join(os.path.dirname(HERE), "include", "bt_hip.h")]
EXPORTS = os.path.join(CSRC, "exports.map")   # linker version script: bt_* global, everything else local
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, debug: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)
    flavours = [("", [], LIB)] + ([("_dbg", ["-DBT_DEBUG_HOOKS"], LIB_DBG)] if debug else [])
    jobs = []
    for suffix, extra, _ in flavours:
        for src in SOURCES:
            s = os.path.join(CSRC, src)
            o = os.path.join(BUILD, src.replace(".hip", suffix + ".o"))
            if force or _stale(o, [s] + HEADERS):
                jobs.append((s, o, extra))

    def compile_one(job):
        s, o, extra = job
        cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[bt build] compiled {os.path.basename(o)}", flush=True)

    # the two gemm objects dominate (~50 s each): start them first so they overlap
    jobs.sort(key=lambda j: 0 if "gemm" in j[0] else 1)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    for suffix, _, lib in flavours:
        objs = [os.path.join(BUILD, s.replace(".hip", suffix + ".o")) for s in SOURCES]
        if force or jobs or _stale(lib, objs + [EXPORTS]):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={EXPORTS}", "-o", lib] + objs
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
            if verbose:
                print(f"[bt build] linked {lib}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, debug="--no-debug" not in sys.argv)
