#!/usr/bin/env python
"""Can the HBM-bound global-attention step of one half-batch run UNDER the MFMA-bound GEMMs of the other (SURVEY section 7
"two roofs")?  Two streams: A = 12 global-attention launches (one block-decoder step of a half batch), B = the GEMM chain of
one token-decoder forward of the other half batch; each timed alone and together.  Variants: attention one-workgroup-per-
item vs persistent (k workgroups per CU), GEMM dispatch auto vs forced 128^2 tiles.  Debug library (bt_debug_* hooks)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from block_transformer_amd import _lib, ops  # noqa: E402
from block_transformer_amd._lib import check, lib, ptr  # noqa: E402

_lib.use_debug_library(True)
DEV = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--n", type=int, default=528)
    ap.add_argument("--rows", type=int, default=2048, help="GEMM rows (token-decoder step of the other half batch)")
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
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

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

    def timed(fa, fb):
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        start = torch.cuda.Event(enable_timing=True)
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

    # size the GEMM chain so that alone it takes about as long as the 12 attention launches
    lib().bt_debug_attn_persist(0)
    lib().bt_debug_gemm_variant(0)
    timed(attn, lambda s: gemms(s, 2))
    ta = min(timed(attn, None)[0] for _ in range(args.iters))
    tg1 = min(timed(None, lambda s: gemms(s, 4))[1] for _ in range(args.iters)) / 4
    reps = max(1, round(ta / tg1))
    for gv in (0, 1):
        lib().bt_debug_gemm_variant(gv)
        tg = min(timed(None, lambda s: gemms(s, reps))[1] for _ in range(args.iters))
        for persist in (0, 1, 2, 3, 4):
            lib().bt_debug_attn_persist(persist)
            ta = min(timed(attn, None)[0] for _ in range(args.iters))
            both = [timed(attn, lambda s: gemms(s, reps)) for _ in range(args.iters)]
            tb = min(max(a, b) for a, b in both)
            print(json.dumps({"op": "overlap", "B": B, "n": n, "gemm_rows": M, "gemm_variant": gv, "attn_persist_wg_per_cu": persist,
                              "attn_alone_ms": round(ta, 3), "gemm_alone_ms": round(tg, 3), "gemm_chain_reps": reps,
                              "together_ms": round(tb, 3), "serial_ms": round(ta + tg, 3),
                              "hidden_frac_of_shorter": round((ta + tg - tb) / min(ta, tg), 3)}), flush=True)
    lib().bt_debug_attn_persist(0)
    lib().bt_debug_gemm_variant(0)


if __name__ == "__main__":
    main()
