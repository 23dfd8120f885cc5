#!/usr/bin/env python
"""Register / LDS / spill report of the compiled gfx950 kernels (reads the code object inside csrc/_build/<name>.o).

    python tools/kernel_resources.py gemm [filter-substring]        # product object
    python tools/kernel_resources.py gemm_dbg mfma256pp             # debug object

Used after every kernel edit: spills or a VGPR count that crosses an occupancy step change the schedule the design assumes.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj_path: str, tmp: str) -> str:
    work = os.path.join(tmp, os.path.basename(obj_path))
    shutil.copy(obj_path, work)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", work], capture_output=True, text=True, check=True)
    for f in os.listdir(tmp):
        if "hipv4-amdgcn" in f:
            return os.path.join(tmp, f)
    raise RuntimeError("no gfx950 bundle in " + obj_path)


def demangle(names):
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not filt:
        return names
    r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.strip().split("\n")


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    obj = os.path.join(ROOT, "block_transformer_amd", "csrc", "_build", which + ".o")
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(obj, tmp)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    ents = notes.split("- .agpr_count:")[1:]
    rows = []
    for e in ents:
        def g(k):
            m = re.search(r"\.%s:\s+(\S+)" % k, e)
            return m.group(1) if m else "?"
        rows.append((g("name"), e.split()[0], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                     g("group_segment_fixed_size"), g("private_segment_fixed_size")))
    names = demangle([r[0] for r in rows])
    print(f"{'kernel':110s} agpr vgpr sgpr vspill sspill   lds scratch")
    for (r, n) in zip(rows, names):
        if sub and sub not in n:
            continue
        n = re.sub(r"\(.*", "", n.replace("void bt::", ""))
        print(f"{n[:110]:110s} {r[1]:>4s} {r[2]:>4s} {r[3]:>4s} {r[4]:>6s} {r[5]:>6s} {r[6]:>6s} {r[7]:>6s}")


if __name__ == "__main__":
    main()
