#!/usr/bin/env python
"""Stand-alone kernel probe: times single C-ABI ops at benchmark shapes with HIP events on the launch stream, and is the
target of the rocprofv3 PMC passes (small, deterministic dispatch list).

    python tools/kernel_probe.py gemm  [--m 32768 --n 6144 --k 2048] [--iters 20]
    python tools/kernel_probe.py attn  [--batch 1024 --n 528]           # global decode step, 1.2B head geometry
    python tools/kernel_probe.py flash [--batch 64 --n 512]
    python tools/kernel_probe.py suite                                   # the shapes of block_main_b4_1.2b
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from block_transformer_amd import _lib, ops  # noqa: E402
from block_transformer_amd._lib import check, current_stream, lib, ptr  # noqa: E402

# the A/B and ablation variants (bt_debug_*) exist only in the debug build; --product times libbt_hip.so itself
PRODUCT = "--product" in sys.argv
if PRODUCT:
    sys.argv.remove("--product")
else:
    _lib.use_debug_library(True)


class _Hooks:
    """bt_debug_* setters; no-ops on the product library (which has none: its dispatch is fixed)."""

    def __getattr__(self, name):
        if name.startswith("bt_debug_") and PRODUCT:
            return lambda *a: None
        return getattr(lib(), name)


hooks = _Hooks()

DEV = "cuda"


def timed(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters  # ms


def probe_gemm(M, N, K, iters, act="none", bias=True, res=False, scratch=False, pad_a=0):
    x = (torch.randn(M, K + pad_a, device=DEV) * 1.0).bfloat16()[:, :K]   # pad_a: row pitch of A = K + pad_a elements
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=DEV).bfloat16() if bias else None
    r = torch.randn(M, N, device=DEV).bfloat16() if res else None
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    sc = torch.empty(16 * M * N, dtype=torch.float32, device=DEV) if scratch else None
    ms = timed(lambda: ops.linear(x, w, b, r, act=act, out=out, scratch=sc), iters)
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    return {"op": "gemm", "M": M, "N": N, "K": K, "act": act, "ms": round(ms, 4), "TFLOPs": round(tf, 1),
            "frac_mfma_peak": round(tf / 2500.0, 3), "splitk_scratch": scratch,
            "weight_GBps": round(N * K * 2 / (ms * 1e-3) / 1e9, 1)}


def probe_lin2(M, N, K1, K2, iters):
    """attention.dense (K2) + mlp.dense_4h_to_h (K1) onto the residual stream: one bt_linear2 GEMM vs the two bt_linear launches."""
    x1 = torch.randn(M, K1, device=DEV).bfloat16()
    x2 = torch.randn(M, K2, device=DEV).bfloat16()
    w1 = (torch.randn(N, K1, device=DEV) * K1 ** -0.5).bfloat16()
    w2 = (torch.randn(N, K2, device=DEV) * K2 ** -0.5).bfloat16()
    b1, b2 = torch.randn(N, device=DEV).bfloat16(), torch.randn(N, device=DEV).bfloat16()
    x = torch.randn(M, N, device=DEV).bfloat16()
    st = current_stream()

    def one():
        check(lib().bt_linear2(_lib.BT_BF16, M, N, K1, ptr(x1), K1, ptr(w1), ptr(b1), K2, ptr(x2), K2, ptr(w2), ptr(b2), ptr(x), N,
                               ptr(x), N, st))

    def two():
        ops.linear(x2, w2, b2, x, out=x)
        ops.linear(x1, w1, b1, x, out=x)

    out = {"op": "lin2", "M": M, "N": N, "K1": K1, "K2": K2}
    for name, fn in (("two_ms", two), ("one_ms", one), ("two_ms_b", two), ("one_ms_b", one)):
        out[name] = round(timed(fn, iters), 4)
    out["TFLOPs_one"] = round(2.0 * M * N * (K1 + K2) / (min(out["one_ms"], out["one_ms_b"]) * 1e-3) / 1e12, 1)
    out["TFLOPs_two"] = round(2.0 * M * N * (K1 + K2) / (min(out["two_ms"], out["two_ms_b"]) * 1e-3) / 1e12, 1)
    return out


def probe_sk(M, iters, reps=3):
    """Round 5: the decode step's GEMMs at M rows, stream-K items kernel (bt_debug_stream_k 1) vs the round-4 dispatch (0), interleaved
    on one box, best of `reps`: the fused QKV + RoPE + append projection (N = 3H), and the two residual-stream projections -- ONE
    stream-K GEMM over K = I + H (bt_linear2_ws) vs attention.dense + dense_4h_to_h as two launches (256 x 128 one-tile kernel at
    M = 4096).  1.2B geometry (H = 2048, I = 8192, 16 heads x 128)."""
    H, I, heads, d = 2048, 8192, 16, 128
    B = M
    x = torch.randn(M, H, device=DEV).bfloat16()
    wq = (torch.randn(3 * H, H, device=DEV) * H ** -0.5).bfloat16()
    bq = torch.randn(3 * H, device=DEV).bfloat16()
    qkv = torch.empty(M, 3 * H, device=DEV, dtype=torch.bfloat16)
    n_max = 8
    kc = torch.zeros(B, heads, n_max, d, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    tab = ops.rope_tables(64, 32, 10000.0, DEV)
    att = torch.randn(M, H, device=DEV).bfloat16()
    h4 = torch.randn(M, I, device=DEV).bfloat16()
    wd = (torch.randn(H, H, device=DEV) * H ** -0.5).bfloat16()
    w2 = (torch.randn(H, I, device=DEV) * I ** -0.5).bfloat16()
    bd, b2 = torch.randn(H, device=DEV).bfloat16(), torch.randn(H, device=DEV).bfloat16()
    xres = torch.randn(M, H, device=DEV).bfloat16()
    scratch = ops.stream_k_scratch(DEV)
    st = current_stream()

    def qkv_call():
        check(lib().bt_qkv_rope_append(_lib.BT_BF16, B, 1, heads, d, 32, ptr(x), H, ptr(wq), ptr(bq), ptr(qkv), 3 * H, ptr(kc), ptr(vc),
                                       n_max, None, 3, None, ptr(tab[0]), ptr(tab[1]), ptr(scratch), scratch.numel(), st))

    def lin2_one():
        ops.linear2(h4, w2, b2, att, wd, bd, residual=xres, out=xres, scratch=scratch)

    def lin2_two():
        ops.linear(att, wd, bd, xres, out=xres)
        ops.linear(h4, w2, b2, xres, out=xres)

    best = {}
    for _ in range(reps):
        for sk in (2, 0):          # 2 = stream-K also where the persistent kernel would take the shape (M = 4096 QKV)
            hooks.bt_debug_stream_k(sk)
            for name, fn in (("qkv", qkv_call), ("res_one_launch", lin2_one), ("res_two_launches", lin2_two)):
                if name == "res_one_launch" and sk == 0:
                    continue          # without stream-K bt_linear2_ws is the two launches again
                ms = timed(fn, iters)
                key = (name, sk)
                best[key] = min(best.get(key, 1e9), ms)
    hooks.bt_debug_stream_k(1)
    fl_qkv = 2.0 * M * 3 * H * H
    fl_res = 2.0 * M * H * (H + I)
    rec = {"op": "stream_k_decode_shapes", "M": M,
           "qkv_us": {"stream_k": round(best[("qkv", 2)] * 1e3, 2), "round4": round(best[("qkv", 0)] * 1e3, 2)},
           "qkv_TFLOPs": {"stream_k": round(fl_qkv / best[("qkv", 2)] / 1e9, 1), "round4": round(fl_qkv / best[("qkv", 0)] / 1e9, 1)},
           "res_us": {"stream_k_one_launch": round(best[("res_one_launch", 2)] * 1e3, 2),
                      "two_launches": round(min(best[("res_two_launches", 0)], best[("res_two_launches", 2)]) * 1e3, 2)},
           "res_TFLOPs": {"stream_k_one_launch": round(fl_res / best[("res_one_launch", 2)] / 1e9, 1),
                          "two_launches": round(fl_res / min(best[("res_two_launches", 0)], best[("res_two_launches", 2)]) / 1e9, 1)}}
    return rec


def probe_attn(B, n, iters, heads=16, d=128, n_split=1, dtype=torch.bfloat16):
    H = heads * d
    n_max = n + 1
    qkv = torch.randn(B, 3 * H, device=DEV).to(dtype)
    kc = torch.randn(B, heads, n_max, d, device=DEV).to(dtype)
    vc = torch.randn(B, heads, n_max, d, device=DEV).to(dtype)
    out = torch.empty(B, H, device=DEV, dtype=dtype)
    pos = torch.tensor([n - 1], dtype=torch.int32, device=DEV)
    ws = torch.empty(B * heads * max(n_split, 1) * (d + 2), dtype=torch.float32, device=DEV)

    def run():
        check(lib().bt_attention(_lib.dtype_code(dtype), 0, B, 1, heads, d, ptr(qkv), 3 * H, ptr(kc), ptr(vc), n_max,
                                 ptr(pos), 0, None, 0, ptr(out), H, n_split, ptr(ws), ws.numel() * 4, current_stream()))

    ms = timed(run, iters)
    nbytes = 2.0 * B * heads * n * d * qkv.element_size()
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"op": "attn_rows_global", "B": B, "n_keys": n, "n_split": n_split, "ms": round(ms, 4),
            "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / 8000.0, 3)}


def probe_local(B, q_len, pos0, iters, heads=16, d=128, window=5):
    """Token-decoder local attention (attend_rows_kernel<..., NW = 1>): q_len new rows at cache rows pos0.., window of E+L-1."""
    H = heads * d
    qkv = torch.randn(B * q_len, 3 * H, device=DEV).bfloat16()
    kc = torch.randn(B, heads, window, d, device=DEV).bfloat16()
    vc = torch.randn(B, heads, window, d, device=DEV).bfloat16()
    out = torch.empty(B * q_len, H, device=DEV, dtype=torch.bfloat16)

    def run():
        check(lib().bt_attention(_lib.BT_BF16, 0, B, q_len, heads, d, ptr(qkv), 3 * H, ptr(kc), ptr(vc), window, None, pos0,
                                 None, 0, ptr(out), H, 1, None, 0, current_stream()))

    ms = timed(run, iters)
    keys = sum(pos0 + t + 1 for t in range(q_len))
    nbytes = 2.0 * B * heads * keys * d * 2 + 2.0 * B * q_len * H * 2     # K,V rows read + q read + out written
    return {"op": "attn_rows_local", "B": B, "q_len": q_len, "pos0": pos0, "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1),
            "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1)}


def probe_flash(B, n, iters, heads=16, d=128):
    H = heads * d
    qkv = torch.randn(B * n, 3 * H, device=DEV).bfloat16()
    kc = torch.randn(B, heads, n, d, device=DEV).bfloat16()
    vc = torch.randn(B, heads, n, d, device=DEV).bfloat16()
    out = torch.empty(B * n, H, device=DEV, dtype=torch.bfloat16)

    def run():
        check(lib().bt_attention(_lib.BT_BF16, 1, B, n, heads, d, ptr(qkv), 3 * H, ptr(kc), ptr(vc), n, None, 0, None, 0,
                                 ptr(out), H, 1, None, 0, current_stream()))

    ms = timed(run, iters)
    fl = 4.0 * B * heads * d * (n * (n + 1) / 2)
    tf = fl / (ms * 1e-3) / 1e12
    return {"op": "attn_flash_prefill", "B": B, "n": n, "ms": round(ms, 4), "TFLOPs": round(tf, 1),
            "frac_mfma_peak": round(tf / 2500.0, 3)}


def probe_copy(nbytes, iters):
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=DEV).normal_()
    b = torch.empty_like(a)
    ms = timed(lambda: b.copy_(a), iters)
    return {"op": "copy_ceiling", "MB": nbytes / 1e6, "ms": round(ms, 4), "GBps": round(2 * nbytes / (ms * 1e-3) / 1e9, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["gemm", "attn", "flash", "suite", "copy", "ab", "abl", "skinny", "decode", "vendor", "pitch",
                                     "local", "lin2", "ln", "ppabl", "ppstore", "ppv2", "ppepi", "ppab", "t21", "ppqkv", "sk", "pptl"])
    ap.add_argument("--m", type=int, default=32768)
    ap.add_argument("--n", type=int, default=6144)
    ap.add_argument("--k", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--keys", type=int, default=528)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--split", type=int, default=1)
    ap.add_argument("--ab-other", type=int, default=9, help="GEMM variant compared with production in `ab`")
    ap.add_argument("--variant", type=int, default=0, help="bt_debug_gemm_variant: 0 auto, 1 128^2 kernel, 2 256^2 kernel")
    ap.add_argument("--act", default="none", choices=["none", "gelu"], help="`gemm`: epilogue activation")
    ap.add_argument("--res", action="store_true", help="`gemm`: residual epilogue")
    args = ap.parse_args()
    res = []
    hooks.bt_debug_gemm_variant(args.variant)
    if args.what == "gemm":
        res.append(probe_gemm(args.m, args.n, args.k, args.iters, act=args.act, res=args.res))
    elif args.what == "sk":      # round 5: stream-K items kernel vs the round-4 dispatch on the decode step's GEMM shapes
        for M in ((args.m,) if args.m != 32768 else (4096, 8192, 2048, 6144, 32768)):
            res.append(probe_sk(M, args.iters if M < 32768 else 10))
    elif args.what == "lin2":    # the residual-stream projections of block_main_b4_1.2b at decode / prefill stack heights
        for M in (4096, 8192, 16384, 32768):
            res.append(probe_lin2(M, 2048, 8192, 2048, args.iters))
    elif args.what == "local":   # the four local steps of a token block at batch --batch: step 0 (q = E = 2), then rows 2, 3, 4
        res.append(probe_local(args.batch, 2, 0, args.iters))
        for pos0 in (2, 3, 4):
            res.append(probe_local(args.batch, 1, pos0, args.iters))
    elif args.what == "attn":
        for nt in ((0,) if PRODUCT else (0, 1, 0, 1)):
            hooks.bt_debug_attn_nt(nt)
            r = probe_attn(args.batch, args.keys, args.iters, n_split=args.split)
            r["nontemporal"] = nt
            res.append(r)
        hooks.bt_debug_attn_nt(0)
    elif args.what == "flash":
        names = {0: "production", 2: "v2 <8 waves, 4-deep ring>", 1: "v1", 3: "OPT 0 (round-1 kernel)", 4: "OPT 1 (K/V DMA before Q)",
                 5: "OPT 3 (+ heaviest block first)", 6: "OPT 7 (+ setprio)", 7: "OPT 5 (DMA first + setprio)"}
        outs = {}
        names[8] = "8 waves x 16 query rows, 4 waves per SIMD"
        names[5] = "OPT 3 (DMA first + heaviest first; O accumulated row-major)"
        names[7] = "OPT 15 (transposed O + setprio)"
        names[0] = "production (OPT 11: transposed O accumulators)"
        names.update({9: "OPT 31 (+ scale in the exponent FMA)", 10: "OPT 63 (+ lazy rescale)", 11: "OPT 59 (63 without setprio)"})
        names[12] = "OPT 95 (31 + XCD-aware grid: a sequence-head's query blocks share one L2)"
        names[0] = "production (v3: persistent workgroups)"
        names[13] = "v2 OPT 95 (one workgroup per query block)"
        for v in ((0,) if PRODUCT else (9, 13, 0, 9, 13, 0) if args.split == 1 else (3, 4, 5, 6, 7, 0, 3, 4, 5, 6, 7, 0)):
            hooks.bt_debug_flash_variant(v)
            r = probe_flash(args.batch, args.keys, args.iters)
            r["variant"] = names[v]
            res.append(r)
        hooks.bt_debug_flash_variant(0)
    elif args.what == "ln":      # dual LayerNorm at H = 2048: register-resident kernel vs the three-pass streaming kernel
        for M in (4096, 32768, 131072):
            x = torch.randn(M, 2048, device=DEV).bfloat16()
            w = [torch.randn(2048, device=DEV).bfloat16() for _ in range(4)]
            o1, o2 = torch.empty_like(x), torch.empty_like(x)
            st = current_stream()
            rec = {"op": "layernorm_dual", "M": M, "H": 2048}
            for tag, flag in (("streaming", 1), ("register", 0), ("streaming_b", 1), ("register_b", 0)):
                hooks.bt_debug_ln_streaming(flag)
                ms = timed(lambda: check(lib().bt_layernorm(_lib.BT_BF16, M, 2048, ptr(x), 2048, ptr(w[0]), ptr(w[1]), ptr(o1), 2048,
                                                            ptr(w[2]), ptr(w[3]), ptr(o2), 2048, 1e-5, st)), args.iters)
                rec[tag + "_us"] = round(ms * 1e3, 2)
                rec[tag + "_GBps"] = round(3 * M * 2048 * 2 / (ms * 1e-3) / 1e9, 1)
            hooks.bt_debug_ln_streaming(0)
            res.append(rec)
    elif args.what == "copy":
        res.append(probe_copy(1 << 30, args.iters))
    elif args.what == "ab":
        # interleaved A/B of the two MFMA kernels on the prefill / large-batch decode shapes
        for (M, N, K) in ((32768, 6144, 2048), (32768, 2048, 2048), (32768, 8192, 2048), (32768, 2048, 8192),
                          (4096, 8192, 2048), (16384, 2048, 2048), (8192, 8192, 2048)):
            for v in (args.ab_other, 0, args.ab_other, 0):   # e.g. 9 = one tile per workgroup, 14 = 32x32x16 MFMA; 0 = production
                hooks.bt_debug_gemm_variant(v)
                for kw in ({}, {"act": "gelu"}, {"res": True}):
                    r = probe_gemm(M, N, K, 10 if M > 8000 else 20, **kw)
                    r["variant"] = v
                    r["res"] = bool(kw.get("res"))
                    res.append(r)
        hooks.bt_debug_gemm_variant(0)
    elif args.what == "decode":
        # decode-regime GEMMs (M = batch): dispatch heuristic (0) vs forced persistent 256^2 (12) vs forced 128^2 (1)
        for (M, N, K) in ((4096, 6144, 2048), (4096, 2048, 2048), (4096, 8192, 2048), (4096, 2048, 8192),
                          (4096, 50304, 2048), (2048, 6144, 2048), (2048, 8192, 2048), (1024, 8192, 2048)):
            for v in (0, 12, 1, 0, 12, 1):
                hooks.bt_debug_gemm_variant(v)
                r = probe_gemm(M, N, K, 20, bias=(N != 50304))
                r["variant"] = v
                res.append(r)
        hooks.bt_debug_gemm_variant(0)
    elif args.what == "vendor":
        # calibration: this library's bt_linear vs torch's bf16 F.linear (hipBLASLt / rocBLAS under PyTorch-ROCm), same box,
        # same inputs, interleaved.  Not part of the product path -- a yardstick for the MFMA roofline fraction.
        import torch.nn.functional as F
        for (M, N, K) in ((32768, 6144, 2048), (32768, 2048, 2048), (32768, 8192, 2048), (32768, 2048, 8192),
                          (4096, 6144, 2048), (4096, 2048, 2048), (4096, 8192, 2048), (4096, 2048, 8192), (4096, 50304, 2048)):
            x = torch.randn(M, K, device=DEV).bfloat16()
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
            b = torch.randn(N, device=DEV).bfloat16()
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            it = 10 if M > 8000 else 20
            for rep in range(2):
                ms_bt = timed(lambda: ops.linear(x, w, b, out=out), it)
                ms_v = timed(lambda: F.linear(x, w, b), it)
                fl = 2.0 * M * N * K
                res.append({"op": "gemm_vs_vendor", "M": M, "N": N, "K": K, "bt_linear_TFLOPs": round(fl / ms_bt / 1e9, 1),
                            "torch_F_linear_TFLOPs": round(fl / ms_v / 1e9, 1), "ratio": round(ms_v / ms_bt, 3)})
            ref = F.linear(x, w, b).float()
            err = float((out.float() - ref).abs().max() / ref.abs().max())
            res[-1]["max_rel_diff_vs_vendor"] = err
    elif args.what == "pitch":
        # partition-camping check: A rows at a power-of-two pitch (K * 2 bytes) vs padded pitches
        for (M, N, K) in ((32768, 2048, 8192), (32768, 8192, 2048), (32768, 6144, 2048)):
            for pad in (0, 64, 128, 0, 64, 128):
                r = probe_gemm(M, N, K, 10, pad_a=pad)
                r["pad_a"] = pad
                res.append(r)
    elif args.what == "skinny":
        for M in (1, 8, 64, 128, 256):
            for (N, K) in ((6144, 2048), (2048, 2048), (8192, 2048), (2048, 8192), (4096, 2048), (50304, 2048)):
                for sc in (False, True):
                    res.append(probe_gemm(M, N, K, 30, scratch=sc))
                if M <= 8:   # GEMV off: the split-K MFMA path it replaces
                    hooks.bt_debug_gemm_variant(22)
                    r = probe_gemm(M, N, K, 30, scratch=True)
                    r["variant"] = "no_gemv"
                    res.append(r)
                    hooks.bt_debug_gemm_variant(0)
    elif args.what == "ppabl":
        # timing ablations of the PRODUCTION persistent kernel (plain epilogue; results of the ablated variants are wrong by design):
        # where do the cycles between the MFMA skeleton and the full kernel go?  variant = 30 + ABL bits
        # (1 no DMA, 2 no ds_reads, 4 no epilogue, 8 every K-tile reads K-tile 0 = all L2 hits, 16 no MFMA, 32 K-order rotation)
        names = {0: "production", 31: "no DMA", 32: "no ds_reads", 33: "MFMA + barriers + epilogue", 34: "no epilogue",
                 37: "MFMA + barriers only", 38: "operand loads all hit L2", 42: "L2 hits, no epilogue", 46: "no MFMA",
                 48: "DMA ring + epilogue", 52: "DMA ring only", 60: "DMA ring only, L2 hits", 62: "K-order rotation",
                 66: "K-order rotation, no epilogue", 84: "K-order rotation, DMA ring only"}
        order = [0, 62, 38, 34, 66, 42, 31, 32, 37, 46, 52, 84, 60]
        for (M, N, K) in ((32768, 6144, 2048), (32768, 2048, 8192), (32768, 8192, 2048), (32768, 2048, 2048), (8192, 6144, 2048)):
            x = torch.randn(M, K, device=DEV).bfloat16()
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
            b = torch.randn(N, device=DEV).bfloat16()
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            for rep in range(2):
                for v in order:
                    hooks.bt_debug_gemm_variant(v)
                    ms = timed(lambda: ops.linear(x, w, b, out=out), 10)
                    res.append({"op": "pp_ablation", "M": M, "N": N, "K": K, "variant": v, "what": names[v], "us": round(ms * 1e3, 1),
                                "TF_equiv": round(2.0 * M * N * K / ms / 1e9, 1)})
            hooks.bt_debug_gemm_variant(62)          # the rotated K order is a correct kernel: check it
            ops.linear(x, w, b, out=out)
            hooks.bt_debug_gemm_variant(0)
            ref = torch.empty_like(out)
            ops.linear(x, w, b, out=ref)
            res.append({"op": "pp_krot_check", "M": M, "N": N, "K": K,
                        "max_abs_diff_vs_production": float((out.float() - ref.float()).abs().max()),
                        "max_abs": float(ref.float().abs().max())})
        hooks.bt_debug_gemm_variant(0)
    elif args.what == "ppstore":
        # where does the epilogue's cost come from?  (profiles/r3_pp_ablation.md: removing the epilogue saves 10-17 % although
        # its instructions are few): dbg 1 = the same stores into an L2-resident dummy region (no HBM write burst), 2 =
        # non-temporal stores, 0xN00 = start-up stagger of the A-panel groups by N us, variant 34 = no epilogue at all
        cases = [("production", 0, 0), ("no epilogue", 34, 0), ("stores to L2-resident dummy", 0, 1), ("nt stores", 0, 2),
                 ("stagger 1 us", 0, 0x100), ("stagger 2 us", 0, 0x200), ("stagger 4 us", 0, 0x400), ("stagger 2 us + nt", 0, 0x202)]
        for (M, N, K) in ((32768, 6144, 2048), (32768, 2048, 8192), (32768, 8192, 2048), (32768, 2048, 2048), (8192, 6144, 2048),
                          (4096, 8192, 2048)):
            x = torch.randn(M, K, device=DEV).bfloat16()
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
            b = torch.randn(N, device=DEV).bfloat16()
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            for rep in range(3):
                for (name, v, dbg) in cases:
                    hooks.bt_debug_gemm_variant(v)
                    hooks.bt_debug_pp_dbg(dbg)
                    ms = timed(lambda: ops.linear(x, w, b, out=out), 10)
                    res.append({"op": "pp_store_path", "M": M, "N": N, "K": K, "case": name, "us": round(ms * 1e3, 1),
                                "TF_equiv": round(2.0 * M * N * K / ms / 1e9, 1)})
            hooks.bt_debug_gemm_variant(0)
            hooks.bt_debug_pp_dbg(0x200)         # staggered start is a correct kernel: check it
            ops.linear(x, w, b, out=out)
            hooks.bt_debug_pp_dbg(0)
            ref = torch.empty_like(out)
            ops.linear(x, w, b, out=ref)
            res.append({"op": "pp_stagger_check", "M": M, "N": N, "K": K, "bit_identical": bool(torch.equal(out, ref))})
    elif args.what == "ppv2":
        # second-generation K-loop of the persistent kernel (variant 0 = production) vs the first generation (16), interleaved,
        # every epilogue mode; bit-identity of the two; timing ablations of the new kernel (100 + ABL bits, plain mode)
        shapes = ((32768, 6144, 2048), (32768, 2048, 8192), (32768, 8192, 2048), (32768, 2048, 2048), (8192, 6144, 2048),
                  (8192, 2048, 8192), (4096, 8192, 2048), (4096, 6144, 2048), (4096, 50304, 2048))
        for (M, N, K) in shapes:
            x = torch.randn(M, K, device=DEV).bfloat16()
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
            b = torch.randn(N, device=DEV).bfloat16()
            r = torch.randn(M, N, device=DEV).bfloat16()
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            modes = (("plain", {}), ("gelu", {"act": "gelu"}), ("res", {"residual": r})) if N != 50304 else (("plain", {}),)
            for name, kw in modes:
                rec = {"op": "pp_v1_vs_v2", "M": M, "N": N, "K": K, "mode": name}
                outs = {}
                for rep in range(3):
                    for v in (16, 12):     # 12 = force the persistent kernel family (production K-loop), 16 = first generation
                        hooks.bt_debug_gemm_variant(v)
                        ms = timed(lambda: ops.linear(x, w, b, kw.get("residual"), act=kw.get("act", "none"), out=out), 10)
                        rec.setdefault("v1_us" if v == 16 else "v2_us", []).append(round(ms * 1e3, 1))
                        outs[v] = out.clone()
                rec["bit_identical"] = bool(torch.equal(outs[16], outs[12]))
                rec["v2_TF"] = round(2.0 * M * N * K / min(rec["v2_us"]) / 1e6, 1)
                rec["v1_TF"] = round(2.0 * M * N * K / min(rec["v1_us"]) / 1e6, 1)
                res.append(rec)
            if (M, N, K) in shapes[:3]:
                for v, what in ((101, "no DMA"), (102, "no ds_reads"), (104, "no epilogue"), (107, "MFMA + barriers only"),
                                (122, "DMA ring only")):
                    hooks.bt_debug_gemm_variant(v)
                    ms = min(timed(lambda: ops.linear(x, w, b, out=out), 10) for _ in range(2))
                    res.append({"op": "pp2_ablation", "M": M, "N": N, "K": K, "what": what, "us": round(ms * 1e3, 1),
                                "TF_equiv": round(2.0 * M * N * K / ms / 1e9, 1)})
            hooks.bt_debug_gemm_variant(0)
    elif args.what == "ppepi":
        # what the per-tile epilogue of the persistent kernel costs, piece by piece (plain mode; ablated variants are wrong by design)
        cases = [(12, "production"), (104, "no epilogue"), (132, "epilogue without global stores"), (164, "epilogue without LDS transposes"),
                 (196, "epilogue = convert only"), (228, "no accumulator init from the bias row"), (101, "no DMA"), (107, "MFMA + barriers only"),
                 (356, "half of the stores (conservative waits)"), (612, "2 of 16 stores (conservative waits)")]
        if args.split == 2:
            cases = [(12, "production"), (104, "no epilogue"), (132, "epilogue without global stores"),
                     (356, "half of the stores (conservative waits)"), (612, "2 of 16 stores (conservative waits)")]
        for (M, N, K) in ((32768, 6144, 2048), (32768, 8192, 2048), (32768, 2048, 2048), (32768, 2048, 8192), (4096, 8192, 2048)):
            x = torch.randn(M, K, device=DEV).bfloat16()
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
            b = torch.randn(N, device=DEV).bfloat16()
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            for rep in range(3):
                for v, what in cases:
                    hooks.bt_debug_gemm_variant(v)
                    ms = timed(lambda: ops.linear(x, w, b, out=out), 10)
                    res.append({"op": "pp2_epilogue_parts", "M": M, "N": N, "K": K, "what": what, "us": round(ms * 1e3, 1)})
            hooks.bt_debug_gemm_variant(0)
    elif args.what == "ppab":
        # interleaved A/B of persistent-kernel variants (--variant A, --ab-other B) on the bench shapes, every epilogue mode
        va, vb = (args.variant or 12), args.ab_other
        for (M, N, K) in ((32768, 6144, 2048), (32768, 2048, 8192), (32768, 8192, 2048), (32768, 2048, 2048), (8192, 6144, 2048),
                          (8192, 2048, 8192), (4096, 8192, 2048), (4096, 6144, 2048), (4096, 50304, 2048)):
            x = torch.randn(M, K, device=DEV).bfloat16()
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
            b = torch.randn(N, device=DEV).bfloat16()
            r = torch.randn(M, N, device=DEV).bfloat16()
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            modes = (("plain", {}), ("gelu", {"act": "gelu"}), ("res", {"residual": r})) if N != 50304 else (("plain", {}),)
            for name, kw in modes:
                rec = {"op": "pp_ab", "M": M, "N": N, "K": K, "mode": name, "a": va, "b": vb, "a_us": [], "b_us": []}
                outs = {}
                for rep in range(3):
                    for v in (va, vb):
                        hooks.bt_debug_gemm_variant(v)
                        ms = timed(lambda: ops.linear(x, w, b, kw.get("residual"), act=kw.get("act", "none"), out=out), 10)
                        rec["a_us" if v == va else "b_us"].append(round(ms * 1e3, 1))
                        outs[v] = out.clone()
                rec["bit_identical"] = bool(torch.equal(outs[va], outs[vb]))
                rec["a_TF"] = round(2.0 * M * N * K / min(rec["a_us"]) / 1e6, 1)
                rec["b_TF"] = round(2.0 * M * N * K / min(rec["b_us"]) / 1e6, 1)
                res.append(rec)
        hooks.bt_debug_gemm_variant(0)
    elif args.what == "ppqkv":
        # the fused QKV + RoPE + KV-append mode of the persistent kernel (the single largest consumer of a generate()) at the prefill
        # chunk shape (64 sequences x 512 positions) and the decode shapes (4096 x 1, 4096 x 2): interleaved A/B of variants
        heads, d, rot = 16, 128, 32
        H = heads * d
        tab = ops.rope_tables(1024, rot, 10000.0, DEV)
        va, vb = (args.variant or 12), args.ab_other
        for (B, q) in ((64, 512), (4096, 1), (4096, 2)):
            n_max = q + 2
            x = torch.randn(B * q, H, device=DEV).bfloat16()
            w = (torch.randn(3 * H, H, device=DEV) * H ** -0.5).bfloat16()
            b = torch.randn(3 * H, device=DEV).bfloat16()
            qkv = torch.empty(B * q, 3 * H, dtype=torch.bfloat16, device=DEV)
            kc = torch.zeros(B, heads, n_max, d, dtype=torch.bfloat16, device=DEV)
            vc = torch.zeros_like(kc)

            def call():
                check(lib().bt_qkv_rope_append(_lib.BT_BF16, B, q, heads, d, rot, ptr(x), H, ptr(w), ptr(b), ptr(qkv), 3 * H, ptr(kc),
                                               ptr(vc), n_max, None, 0, None, ptr(tab[0]), ptr(tab[1]), None, 0, current_stream()))
            rec = {"op": "pp_qkv_ab", "B": B, "q_len": q, "a": va, "b": vb, "a_us": [], "b_us": []}
            outs = {}
            for rep in range(3):
                for v in (va, vb):
                    hooks.bt_debug_gemm_variant(v)
                    ms = timed(call, 10 if B * q > 8192 else 20)
                    rec["a_us" if v == va else "b_us"].append(round(ms * 1e3, 1))
                    outs[v] = (qkv.view(B * q, heads, 3, d)[:, :, 0].clone(), kc.clone())
            rec["bit_identical"] = bool(torch.equal(outs[va][0], outs[vb][0]) and torch.equal(outs[va][1], outs[vb][1]))
            fl = 2.0 * B * q * 3 * H * H / 1e6
            rec["a_TF"], rec["b_TF"] = round(fl / min(rec["a_us"]), 1), round(fl / min(rec["b_us"]), 1)
            res.append(rec)
        hooks.bt_debug_gemm_variant(0)
    elif args.what == "pptl":
        # round 6: the tile-boundary TIMELINE of the persistent kernel (debug variant 28: every wave stamps s_memtime at ten points of each
        # of its first nine tiles; workgroups 0 .. 3 dump).  Per mode and shape: the median over tiles 1 .. nt - 2 and the 8 waves of the
        # intervals between consecutive stamps, in shader-clock cycles (s_memtime ticks at 100 MHz on this part: converted with the
        # launch's measured duration / its span in ticks) -- where the time between a tile's last MFMA cluster and the next tile's
        # first one goes.  Events: 0 top of phase (0,0), 1 cluster (0,0) opens, 2 its 32 MFMAs issued, 3 cluster (0,1) opens,
        # 4 cluster (1,0) opens, 5 top of phase (nk-1,0), 6 cluster (nk-1,1) opens, 7 issued, 8 epilogue starts (after the boundary
        # barrier + pre-issue), 9 epilogue done (last store issued).
        import statistics
        heads, d, rot = 16, 128, 32
        H = heads * d
        tab = ops.rope_tables(1024, rot, 10000.0, DEV)
        buf = torch.zeros(8 * 8 * 128, dtype=torch.int64, device=DEV)      # first four + last four workgroups x 8 waves x 128 stamps
        hooks.bt_debug_timeline_buffer(ptr(buf))
        # --variant 95: the round-5 tile boundary (the two wave groups' epilogues one after the other); default 28: the round-6 boundary
        tl_variant = 95 if args.variant == 95 else 28
        for (mode, M, N, K) in (("plain", 32768, 6144, 2048), ("gelu", 32768, 8192, 2048), ("qkv", 32768, 6144, 2048), ("res", 32768, 2048, 2048),
                                ("plain", 4096, 8192, 2048), ("gelu", 4096, 8192, 2048), ("qkv", 4096, 6144, 2048), ("qkv2", 8192, 6144, 2048)):
            x = torch.randn(M, K, device=DEV).bfloat16()
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
            b = torch.randn(N, device=DEV).bfloat16()
            r = torch.randn(M, N, device=DEV).bfloat16() if mode == "res" else None
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            if mode.startswith("qkv"):
                q = 512 if M == 32768 else (2 if mode == "qkv2" else 1)
                B = M // q
                n_max = q + 2
                kc = torch.zeros(B, heads, n_max, d, dtype=torch.bfloat16, device=DEV)
                vc = torch.zeros_like(kc)

                def call():
                    check(lib().bt_qkv_rope_append(_lib.BT_BF16, B, q, heads, d, rot, ptr(x), H, ptr(w), ptr(b), ptr(out), 3 * H, ptr(kc),
                                                   ptr(vc), n_max, None, 0, None, ptr(tab[0]), ptr(tab[1]), None, 0, current_stream()))
            else:
                def call():
                    ops.linear(x, w, b, r, act="gelu" if mode == "gelu" else "none", out=out)
            hooks.bt_debug_gemm_variant(0)
            us_prod = timed(call, 10) * 1e3
            hooks.bt_debug_gemm_variant(tl_variant)
            us_tl = timed(call, 10) * 1e3
            buf.zero_()
            call()
            torch.cuda.synchronize()
            hooks.bt_debug_gemm_variant(0)
            raw = buf.view(8, 8, 128).cpu()
            st = raw[:4, :, :117].reshape(4, 8, 9, 13)
            # tiles of workgroups 0 .. 3 (the first of every round); slots beyond hold whatever an earlier launch left in that LDS
            nt = min(9, -(-(-(-M // 256) * -(-N // 256)) // 256))
            span = int(st[:, :, :nt].max() - st[:, :, :nt][st[:, :, :nt] != 0].min())
            rec = {"op": "pp_timeline", "boundary": "round 5 (serial epilogues)" if tl_variant == 95 else "round 6", "mode": mode, "M": M, "N": N, "K": K, "tiles_per_workgroup": nt, "us_production": round(us_prod, 1),
                   "us_with_stamps": round(us_tl, 1), "span_ticks_wg0_3": span, "ns_per_tick_if_span_is_the_launch": round(us_tl * 1e3 / max(span, 1), 3)}
            # kernel entry / exit (slots 117 / 118: s_memtime; 119 / 120: s_memrealtime, 100 MHz) of the first and last four workgroups
            nt_all = -(-(-(-M // 256) * -(-N // 256)) // 256)
            ent, ext, rt0, rt1 = raw[:, :, 117], raw[:, :, 118], raw[:, :, 119], raw[:, :, 120]
            rec["kernel_cycles_median"] = float((ext - ent).float().median())
            rec["fill_cycles_entry_to_first_phase_median"] = float((raw[:4, :, 0] - ent[:4]).float().median())
            if nt_all <= 9:
                last_e9 = raw[:4, :, (nt_all - 1) * 13 + 9]
                rec["tail_cycles_last_epilogue_end_to_exit_median"] = float((ext[:4] - last_e9).float().median())
                rec["last_epilogue_cycles_median"] = float((last_e9 - raw[:4, :, (nt_all - 1) * 13 + 8]).float().median())
            rec["realtime_us"] = {"first_entry_to_last_exit": float(rt1.max() - rt0.min()) * 0.01,
                                  "entry_skew_first4_vs_last4": float(rt0[4:].float().mean() - rt0[:4].float().mean()) * 0.01,
                                  "exit_skew_first4_vs_last4": float(rt1[4:].float().mean() - rt1[:4].float().mean()) * 0.01,
                                  "per_workgroup_resident": float((rt1 - rt0).float().median()) * 0.01}
            names = ["0>1 wait+reads (0,0)", "1>2 cluster (0,0) issue", "2>3 to cluster (0,1)", "3>4 to cluster (1,0)", "4>5 K-tiles 1..nk-2",
                     "5>6 last K-tile to its 2nd cluster", "6>7 last cluster issue", "7>8 boundary barrier + pre-issue", "8>9 epilogue",
                     "9>0' next tile's phase (0,0) top", "7>1' last cluster issued -> next tile's first cluster opens", "1>1' whole tile"]
            iv = {n: [] for n in names}
            # inside the epilogue (events 10 .. 12): sweep A = RoPE / residual loads, set-up of the store addressing, fragments 0-3, 4-7
            epi_parts = {"8>10 epilogue sweep A": (8, 10), "10>11 store set-up": (10, 11), "11>12 fragments 0-3": (11, 12), "12>9 fragments 4-7": (12, 9)}
            per_wave_epi = [[] for _ in range(8)]
            for wg in range(4):
                for wv in range(8):
                    for j in range(1 if nt > 2 else 0, max(nt - 1, 1)):
                        t = [int(v) for v in st[wg, wv, j]]
                        nx = [int(v) for v in st[wg, wv, j + 1]] if j + 1 < nt else None
                        for e in range(9):
                            iv[names[e]].append(t[e + 1] - t[e])
                        per_wave_epi[wv].append(t[9] - t[8])
                        for n, (a, b) in epi_parts.items():
                            iv.setdefault(n, []).append(t[b] - t[a])
                        if nx is not None and nx[0]:
                            iv[names[9]].append(nx[0] - t[9])
                            iv[names[10]].append(nx[1] - t[7])
                            iv[names[11]].append(nx[1] - t[1])
            rec["median_ticks"] = {n: (statistics.median(v) if v else None) for n, v in iv.items()}
            rec["p90_ticks"] = {n: (sorted(v)[int(0.9 * (len(v) - 1))] if v else None) for n, v in iv.items()}
            rec["epilogue_median_ticks_by_wave"] = [statistics.median(v) if v else None for v in per_wave_epi]
            for grp in (0, 1):     # the two wave groups see the boundary differently (who waits for whom)
                g = {n: [] for n in (names[7], names[8], names[0], names[10])}
                for wg in range(4):
                    for wv in range(4 * grp, 4 * grp + 4):
                        for j in range(1 if nt > 2 else 0, max(nt - 1, 1)):
                            t = [int(v) for v in st[wg, wv, j]]
                            g[names[7]].append(t[8] - t[7])
                            g[names[8]].append(t[9] - t[8])
                            if j + 1 < nt:
                                nx = [int(v) for v in st[wg, wv, j + 1]]
                                g[names[0]].append(nx[1] - nx[0])
                                g[names[10]].append(nx[1] - t[7])
                rec[f"group{grp}_median_ticks"] = {n: (statistics.median(v) if v else None) for n, v in g.items()}
            res.append(rec)
        hooks.bt_debug_timeline_buffer(None)
    elif args.what == "t21":
        # the decode projections onto the residual stream at M = 4096: one-tile 256 x 128 kernel (production dispatch, variant 0) vs
        # the 128^2 kernel (variant 1) vs torch F.linear (hipBLASLt), interleaved
        import torch.nn.functional as F
        for (M, N, K) in ((4096, 2048, 2048), (4096, 2048, 8192)):
            x = torch.randn(M, K, device=DEV).bfloat16()
            w = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
            b = torch.randn(N, device=DEV).bfloat16()
            r = torch.randn(M, N, device=DEV).bfloat16()
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            for mode, res_ in (("bias", None), ("bias_res", r)):
                rec = {"op": "t21", "M": M, "N": N, "K": K, "mode": mode}
                for rep in range(3):
                    for name, v, st in (("t256x128_us", 0, 0), ("t256x128_r3_no_lgkm_wait_us", 0, 8), ("t256x128_no_setprio_us", 0, 9),
                                        ("t128x128_us", 1, 0)):
                        hooks.bt_debug_gemm_variant(v)
                        hooks.bt_debug_small_stages(st)     # 9: without s_setprio around its MFMA clusters; 8: round-3 form (SAFE = false)
                        ms = timed(lambda: ops.linear(x, w, b, res_, out=out), 20)
                        rec.setdefault(name, []).append(round(ms * 1e3, 1))
                    hooks.bt_debug_gemm_variant(0)
                    hooks.bt_debug_small_stages(0)
                    ms = timed(lambda: F.linear(x, w, b), 20)
                    rec.setdefault("vendor_us", []).append(round(ms * 1e3, 1))
                fl = 2.0 * M * N * K / 1e6
                rec["TF"] = {k[:-3]: round(fl / min(v), 1) for k, v in rec.items() if k.endswith("_us")}
                res.append(rec)
    elif args.what == "abl":
        for (M, N, K) in ((32768, 6144, 2048), (32768, 2048, 8192)):
            for v in (0, 3, 20, 21, 8, 0, 3, 20, 21, 8):   # production / 4-phase one-tile kernel: full, no MFMA, DMA only, MFMA+barriers only
                hooks.bt_debug_gemm_variant(v)
                r = probe_gemm(M, N, K, 10, bias=False)
                r["variant"] = v
                res.append(r)
        hooks.bt_debug_gemm_variant(0)
    else:
        res.append(probe_copy(1 << 30, 10))
        for M in (32768, 4096, 2048, 1024, 256, 64, 1):
            it = 10 if M >= 4096 else 30
            res.append(probe_gemm(M, 6144, 2048, it))
            res.append(probe_gemm(M, 2048, 2048, it, res=True))
            res.append(probe_gemm(M, 8192, 2048, it, act="gelu"))
            res.append(probe_gemm(M, 2048, 8192, it, res=True))
        for M in (4096, 1024, 256):
            res.append(probe_gemm(M, 50304, 2048, 10, bias=False))
        for B in (4096, 1024, 256, 16, 1):
            res.append(probe_attn(B, 528, 20))
        res.append(probe_attn(1, 528, 20, n_split=16))
        res.append(probe_attn(16, 528, 20, n_split=4))
        res.append(probe_flash(64, 512, 10))
    for r in res:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
