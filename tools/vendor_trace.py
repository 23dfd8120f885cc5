#!/usr/bin/env python
"""Yardstick only (never on the product path): run torch's F.linear (hipBLASLt) on the benchmark GEMM shapes so that a
`rocprofv3 --kernel-trace --stats` around this script shows WHICH vendor kernel (tile / wave-group / prefetch config encoded
in the Tensile kernel name) reaches 1.4-1.57 PF on them."""
import torch
import torch.nn.functional as F

for M, N, K in [(32768, 6144, 2048), (32768, 2048, 8192), (32768, 8192, 2048), (4096, 2048, 2048), (4096, 6144, 2048),
                (2048, 2048, 2048), (2048, 6144, 2048), (2048, 8192, 2048), (2048, 2048, 8192)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        F.linear(a, w, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        F.linear(a, w, b)
    e.record()
    e.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"vendor M={M} N={N} K={K}: {ms * 1e3:.1f} us  {2 * M * N * K / ms / 1e9:.0f} TF", flush=True)
