#!/usr/bin/env python
"""Same-box, same-process A/B of the small- / mid-batch launch structures through the debug library (round 4):
    few_attn  : <= 2 rows, token decoder: local attention inside the residual-stream GEMV (1) vs its own launch (0)
    mid_dual  : 17..64 rows: QKV + h->4h split-K GEMMs in one launch and one reduce (1) vs two each (0)
    mid_attn  : 17..64 rows, token decoder: local attention inside that reduce launch (1) vs its own launch (0)
Prints one JSON line per (batch, knob setting): end-to-end tokens/s of generate(prefill 2048 / decode 128), best of --reps.

    python tools/ab_small_batch.py --knob few_attn --batches 1,2 [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from block_transformer_amd import _lib  # noqa: E402
from block_transformer_amd.config import get_max_length, named_dims  # noqa: E402
from block_transformer_amd.model.utils import load_block_transformer_from_dims  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", choices=["few_attn", "mid_dual", "mid_attn", "few_max4", "mid_tile", "mid_force2", "few_persist", "attn_wide"], required=True)
    ap.add_argument("--batches", default="1,2")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--fixed", type=int, default=None, help="run ONE setting in this process: median / min over --iters generates")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--cutoffs", default="", help="few_max4: compare these few-rows cutoffs (e.g. 16,4,2,1,0); mid_tile: tile variants "
                                                  "(0 = 64x64, 1 = 128x64, 2 = 128x128) instead of on / off")
    ap.add_argument("--prefill", type=int, default=2048)
    ap.add_argument("--decode", type=int, default=128)
    args = ap.parse_args()
    _lib.use_debug_library(True)
    hooks = _lib.lib()
    setter = {"few_attn": hooks.bt_debug_few_attn, "mid_dual": hooks.bt_debug_mid_dual, "mid_attn": hooks.bt_debug_mid_attn,
              # 5..16-row stacks: split-K slab path (on: few-rows cutoff 4) vs the MFMA weight streamer launches (off: cutoff 16)
              "few_max4": lambda on: hooks.bt_debug_few_rows_max(4 if on else 16), "mid_tile": hooks.bt_debug_mid_tile,
              "mid_force2": hooks.bt_debug_mid_force2, "few_persist": hooks.bt_debug_few_persist,
              "attn_wide": hooks.bt_debug_attn_wide}[args.knob]
    dims = named_dims("block_main_b4_1.2b")
    torch.manual_seed(1234)
    with torch.device("cuda"):
        model = load_block_transformer_from_dims(dims, "bf16", keep_unused_embed_in=False)
    model.eval()
    model.token_decoder.generation_config.eos_token_id = None
    ml = get_max_length(dims.L, args.prefill, args.decode)
    if args.fixed is not None:
        # one setting per PROCESS (fresh allocator, fresh engine), many generates, median and min: the in-process interleaved form
        # cannot resolve +-8 % placement effects at ~70 ms per generate
        import statistics
        setter(args.fixed)
        for B in (int(b) for b in args.batches.split(",")):
            ids = torch.randint(0, dims.vocab, (B, args.prefill), generator=torch.Generator().manual_seed(B)).cuda()
            model._engine = None
            for _ in range(3):
                model.generate(ids, max_length=ml)
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.iters):
                t0 = time.perf_counter()
                model.generate(ids, max_length=ml)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            print(json.dumps({"knob": args.knob, "fixed": args.fixed, "batch": B, "iters": args.iters,
                              "median_ms": round(statistics.median(ts) * 1e3, 3), "min_ms": round(min(ts) * 1e3, 3),
                              "tokens_per_s_median": round(B * args.decode / statistics.median(ts), 1)}), flush=True)
            model.release_buffers()
        return
    settings = (1, 0)
    if args.cutoffs:
        settings = tuple(int(c) for c in args.cutoffs.split(","))
        setter = hooks.bt_debug_mid_tile if args.knob == "mid_tile" else hooks.bt_debug_few_rows_max
    for B in (int(b) for b in args.batches.split(",")):
        ids = torch.randint(0, dims.vocab, (B, args.prefill), generator=torch.Generator().manual_seed(B)).cuda()
        best = {v: 1e9 for v in settings}
        for rep in range(args.reps):
            # interleaved, and the order flips every repetition: a generate() of ~70 ms shows a bimodal +-8 % with the placement of the
            # freshly allocated engine, and whichever setting ran first in a pair used to inherit it
            for on in (settings if rep % 2 == 0 else tuple(reversed(settings))):
                setter(on)
                model._engine = None              # captured graphs hold the launch sequence
                model.generate(ids, max_length=ml)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    model.generate(ids, max_length=ml)
                torch.cuda.synchronize()
                best[on] = min(best[on], (time.perf_counter() - t0) / 3)
        setter((0 if args.knob == "mid_tile" else 2) if args.cutoffs else (0 if args.knob in ("few_max4", "mid_tile", "mid_force2") else 1))
        for on in settings:
            print(json.dumps({"knob": args.knob, "on": on, "batch": B, "prefill": args.prefill, "decode": args.decode,
                              "ms_per_generate": round(best[on] * 1e3, 3), "tokens_per_s": round(B * args.decode / best[on], 1)}),
                  flush=True)
        model.release_buffers()


if __name__ == "__main__":
    main()
