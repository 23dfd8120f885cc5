#!/usr/bin/env python
"""Yardstick only (not part of the product): torch's bf16 F.linear (hipBLASLt) on one benchmark shape, a few launches, for a
rocprofv3 --pmc pass -- which clock and MFMA duty cycle does the vendor kernel run at on this box?

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... -- python tools/vendor_pmc_target.py --m 32768 --n 6144 --k 2048
"""
import argparse

import torch
import torch.nn.functional as F

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=32768)
ap.add_argument("--n", type=int, default=6144)
ap.add_argument("--k", type=int, default=2048)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
x = torch.randn(a.m, a.k, device="cuda").bfloat16()
w = (torch.randn(a.n, a.k, device="cuda") * a.k ** -0.5).bfloat16()
b = torch.randn(a.n, device="cuda").bfloat16()
for _ in range(a.iters + 2):
    y = F.linear(x, w, b)
torch.cuda.synchronize()
print(float(y.float().abs().max()))
