#!/usr/bin/env python
"""Eager (no hipGraph) decode steps of a 1.2B-shaped two-layer stack at a mid-batch row count, for `rocprofv3 --pmc` passes over the
fused QKV + h->4h launch (gemm_bf16_mfma256x128_slab_kernel<true, true>) and the residual slab launch -- rocprofv3's counter
collection crashed on bench.py's graph replays (tools/gpu_runs/r6_run24.sh).
    python tools/mid_launch_probe.py --rows 256 --iters 10"""
import argparse
import dataclasses
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from block_transformer_amd.config import named_dims  # noqa: E402
from block_transformer_amd.model.utils import load_block_transformer_from_dims  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--past", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2), td=dataclasses.replace(base.td, layers=2))
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False).bfloat16().to(dev)
    H = dims.bd.hidden
    M = args.rows
    g = torch.Generator().manual_seed(7)
    xp = (torch.randn(M, args.past, H, generator=g) * 0.5).bfloat16().to(dev)
    xs = (torch.randn(M, 1, H, generator=g) * 0.5).bfloat16().to(dev)
    ones = torch.ones(M, args.past + 1, dtype=torch.long, device=dev)
    for _ in range(args.iters):     # (the step appends to the cache in place: a fresh prefill per iteration keeps the shapes fixed)
        pre = model.block_decoder(inputs_embeds=xp, block_attention_mask=ones[:, :args.past])
        model.block_decoder(inputs_embeds=xs, block_attention_mask=ones, past_key_values=pre.past_key_values)
    torch.cuda.synchronize()
    print(f"{args.iters} eager decode steps of {M} rows done")


if __name__ == "__main__":
    main()
