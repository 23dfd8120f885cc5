#!/usr/bin/env python
"""CU-partitioned overlap probe (round 3): can the HBM-bound global-attention step of one half batch run BESIDE the MFMA-bound
GEMM chain of the other half when each stream owns its own compute units?

Round 2 measured 0 % overlap on two ordinary streams: the attention grid fills every CU's wave slots and the persistent GEMM
fills every CU's registers and LDS, so whichever kernel is resident simply excludes the other.  HIP streams can be created
with a CU mask (hipExtStreamCreateWithCUMask): stream A = `--attn-cus` compute units (every k-th CU, so every XCD / shader
engine contributes), stream B = the rest.  Each side is timed alone on ITS partition, alone on the whole chip, and together.

    python tools/cumask_probe.py [--batch 2048 --n 528 --rows 2048]

Debug library (bt_debug_gemm_max_wg caps the persistent GEMM's grid at its partition's CU count).
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from block_transformer_amd import _lib  # noqa: E402
from block_transformer_amd._lib import check, lib, ptr  # noqa: E402

_lib.use_debug_library(True)
DEV = "cuda"
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """torch.cuda.ExternalStream over a HIP stream restricted to the CUs whose bit is set (256 CUs = 8 x uint32)."""
    words = (ctypes.c_uint32 * 8)(*[sum(1 << j for j in range(32) if bits[w * 32 + j]) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(s.value), s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--n", type=int, default=528)
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    B, n, M, H, heads, d = args.batch, args.n, args.rows, 2048, 16, 128
    qkv = torch.randn(B, 3 * H, device=DEV).bfloat16()
    kc = torch.randn(B, heads, n, d, device=DEV).bfloat16()
    vc = torch.randn(B, heads, n, d, device=DEV).bfloat16()
    out = torch.empty(B, H, device=DEV, dtype=torch.bfloat16)
    x = torch.randn(M, H, device=DEV).bfloat16()
    x4 = torch.randn(M, 4 * H, device=DEV).bfloat16()
    shapes = [(3 * H, H), (H, H), (4 * H, H), (H, 4 * H)]
    ws_ = [torch.randn(N, K, device=DEV).bfloat16() * K ** -0.5 for N, K in shapes]
    outs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for N, K in shapes]

    def attn(stream):
        for _ in range(12):
            check(lib().bt_attention(_lib.BT_BF16, 0, B, 1, heads, d, ptr(qkv), 3 * H, ptr(kc), ptr(vc), n, None, n - 1, None,
                                     0, ptr(out), H, 1, None, 0, stream.cuda_stream))

    def gemms(stream, reps):
        for _ in range(reps):
            for (N, K), w, o in zip(shapes, ws_, outs):
                a = x4 if K == 4 * H else x
                check(lib().bt_linear(_lib.BT_BF16, M, N, K, ptr(a), K, ptr(w), None, None, 0, ptr(o), N, 0,
                                      stream.cuda_stream))

    def timed(sa, sb, fa, fb):
        torch.cuda.synchronize()
        start, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        start.record()
        sa.wait_event(start)
        sb.wait_event(start)
        if fa:
            fa(sa)
        e1.record(sa)
        if fb:
            fb(sb)
        e2.record(sb)
        torch.cuda.synchronize()
        return start.elapsed_time(e1), start.elapsed_time(e2)

    full_a, full_b = torch.cuda.Stream(), torch.cuda.Stream()
    lib().bt_debug_gemm_max_wg(0)
    timed(full_a, full_b, attn, lambda s: gemms(s, 2))
    ta_full = min(timed(full_a, full_b, attn, None)[0] for _ in range(args.iters))
    tg1 = min(timed(full_a, full_b, None, lambda s: gemms(s, 4))[1] for _ in range(args.iters)) / 4
    reps = max(1, round(ta_full / tg1))
    tg_full = min(timed(full_a, full_b, None, lambda s: gemms(s, reps))[1] for _ in range(args.iters))
    both_plain = min(max(*timed(full_a, full_b, attn, lambda s: gemms(s, reps))) for _ in range(args.iters))
    print(json.dumps({"op": "cumask_overlap", "partition": "none (two ordinary streams)", "attn_alone_ms": round(ta_full, 3),
                      "gemm_alone_ms": round(tg_full, 3), "gemm_chain_reps": reps, "together_ms": round(both_plain, 3),
                      "serial_ms": round(ta_full + tg_full, 3)}), flush=True)
    for attn_cus, step in ((64, 4), (128, 2), (96, None), (32, 8)):
        if step is not None:
            bits_a = [i % step == 0 for i in range(256)]
        else:       # 96 = 3 of every 8
            bits_a = [i % 8 in (0, 3, 6) for i in range(256)]
        bits_b = [not b for b in bits_a]
        sa, ha = masked_stream(bits_a)
        sb, hb = masked_stream(bits_b)
        n_b = sum(bits_b)
        lib().bt_debug_gemm_max_wg(n_b)
        timed(sa, sb, attn, lambda s: gemms(s, 1))
        ta = min(timed(sa, sb, attn, None)[0] for _ in range(args.iters))
        tg = min(timed(sa, sb, None, lambda s: gemms(s, reps))[1] for _ in range(args.iters))
        both = [timed(sa, sb, attn, lambda s: gemms(s, reps)) for _ in range(args.iters)]
        tb = min(max(a, b) for a, b in both)
        nbytes = 12 * 2.0 * B * heads * n * d * 2
        print(json.dumps({"op": "cumask_overlap", "partition": f"attention {sum(bits_a)} CUs / GEMM {n_b} CUs",
                          "attn_alone_on_partition_ms": round(ta, 3), "attn_GBps_on_partition": round(nbytes / ta / 1e6, 1),
                          "gemm_alone_on_partition_ms": round(tg, 3), "together_ms": round(tb, 3),
                          "attn_ms_together": round(min(a for a, _ in both), 3), "gemm_ms_together": round(min(b for _, b in both), 3),
                          "serial_full_chip_ms": round(ta_full + tg_full, 3),
                          "speedup_vs_serial_full_chip": round((ta_full + tg_full) / tb, 3)}), flush=True)
        lib().bt_debug_gemm_max_wg(0)
        torch.cuda.synchronize()
        hip.hipStreamDestroy(ha)
        hip.hipStreamDestroy(hb)


if __name__ == "__main__":
    main()
