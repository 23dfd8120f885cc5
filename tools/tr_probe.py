#!/usr/bin/env python
"""Empirical semantics of ds_read_b64_tr_b16 on gfx950 (used to design the flash-prefill V operand path)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_transformer_amd import _lib
from block_transformer_amd._lib import check, current_stream, lib, ptr
_lib.use_debug_library(True)   # bt_debug_tr_probe exists only in libbt_hip_dbg.so

def run(addr):
    src = torch.arange(4096, dtype=torch.int16, device="cuda")
    a = torch.tensor(addr, dtype=torch.int32, device="cuda")
    out = torch.zeros(64 * 4, dtype=torch.int16, device="cuda")
    check(lib().bt_debug_tr_probe(ptr(src), ptr(a), ptr(out), current_stream()))
    torch.cuda.synchronize()
    return out.view(64, 4).cpu().tolist()

print("case A: linear addresses, lane l -> element 4*l")
for l, v in enumerate(run([4 * l for l in range(64)])):
    print(l, v)
print("case B: V tile [key][128] row-major, group g: rows kbase+(i>>2), cols (i&3)*4 ; kbase = g*4")
addr = []
for l in range(64):
    g, i = l >> 4, l & 15
    addr.append((g * 4 + (i >> 2)) * 128 + (i & 3) * 4)
for l, v in enumerate(run(addr)):
    print(l, v)
