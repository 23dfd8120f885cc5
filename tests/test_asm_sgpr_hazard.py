"""Inline-asm SGPR hazard scan of the persistent / stream-K GEMM kernels (round 5, profiles/r5_sk_bisect.md).

hipcc pads the "VALU writes SGPR -> VMEM reads that SGPR as base: 5 wait states" hazard for its own instructions but never inside
an `asm` string (CDNA guide 5.7 item 2).  The stream-K kernel's first version read a flag pointer the compiler had just reloaded
with v_readlane_b32 (a spill reload) -- a memory access fault on every launch with a hand-off.  This test compiles gemm.hip to
gfx950 assembly (no GPU needed) and scans every VMEM instruction that takes an SGPR pair as its base (LDS-DMA operand loads, the
sc1 flag accesses) in the production kernels for a v_readlane / v_readfirstlane write of that pair fewer than 5 wait states earlier
in program order."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def gemm_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "gemm.s"
    src = os.path.join(ROOT, "block_transformer_amd", "csrc", "gemm.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(out), src],
                   check=True, capture_output=True, timeout=600)
    return out.read_text()


def _kernels(txt, pattern):
    for k in re.split(r"\n(?=_ZN2bt\d+gemm_bf16_mfma256)", txt):
        m = re.match(pattern, k)
        if m and "s_endpgm" in k:
            yield m.group(0), k[:k.index("s_endpgm")]


def _scan(body):
    ins = [l.strip() for l in body.split("\n")]
    ins = [l for l in ins if l and not l.startswith(";") and not l.startswith(".") and not l.endswith(":")]
    total, hazards = 0, []
    for i, l in enumerate(ins):
        if not (l.startswith("global_load_lds_dwordx4") or (l.startswith("global_") and " sc1" in l)):
            continue
        mm = re.search(r"s\[(\d+):(\d+)\]", l)
        if not mm:
            continue
        total += 1
        lo, hi = int(mm.group(1)), int(mm.group(2))
        states, j = 0, i - 1
        while j >= 0 and states < 5:
            w = re.match(r"v_(readlane|readfirstlane)_b32 s(\d+)", ins[j])
            if w and lo <= int(w.group(2)) <= hi:
                hazards.append((ins[j], l, states))
                break
            n = re.match(r"s_nop (\d+)", ins[j])
            states += (int(n.group(1)) + 1) if n else 1
            j -= 1
    return total, hazards


def test_no_valu_written_sgpr_base_within_five_wait_states(gemm_asm):
    seen = 0
    for name, body in _kernels(gemm_asm, r"_ZN2bt\d+gemm_bf16_mfma256(sk|pp2)_kernelILi\d"):
        total, hazards = _scan(body)
        assert total > 50, (name, total)                 # the scan really found the kernel's DMA / flag instructions
        assert not hazards, (name, hazards[:3])
        seen += 1
    assert seen >= 8, seen                               # four stream-K + five persistent instantiations in the product build


def test_one_tile_and_slab_kernels_have_no_valu_written_sgpr_base(gemm_asm):
    """The 256 x 128 kernels (one tile per workgroup; K ranges into slabs for the mid-batch stacks) issue their LDS-DMA from an SGPR base
    computed from v_readfirstlane(wave): the base must have gone through the scalar ALU (or 5 wait states) before the first DMA."""
    seen = 0
    for name, body in _kernels(gemm_asm, r"_ZN2bt\d+gemm_bf16_mfma256x128_(slab_)?kernel"):
        total, hazards = _scan(body)
        assert total >= 16, (name, total)
        assert not hazards, (name, hazards[:3])
        seen += 1
    assert seen >= 3, seen


def test_the_scan_catches_the_original_bug(gemm_asm):
    """The flag helpers without their leading s_nop 4 are what faulted: strip the nops from the stream-K kernels' text and the scan
    must report the v_readlane -> global_* sc1 pairs."""
    found = 0
    for name, body in _kernels(gemm_asm, r"_ZN2bt\d+gemm_bf16_mfma256sk_kernelILi\d"):
        stripped = re.sub(r"\n\s*s_nop 4\n(\s*v_mov_b32 v\d+, 0\n)", r"\n\1", body)
        _, hazards = _scan(stripped)
        found += len(hazards)
    assert found > 0
