#!/usr/bin/env python
"""Diagnostic (round 6): find the pass in which a stream-K hand-off time-out is flagged.  2-layer block_main_b4_1.2b, B = 4096,
P = 128, D = --decode; runs generate() --reps times (graphs on/off) and after each one reads err_flag and scans the engine's workspaces
for the sticky word (SK_TIMEOUT_MAGIC) and the diagnostics behind it."""
import argparse, dataclasses, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_transformer_amd import _lib
from block_transformer_amd.config import named_dims
from block_transformer_amd.model.utils import load_block_transformer_from_dims

ap = argparse.ArgumentParser()
ap.add_argument("--decode", type=int, default=2048)
ap.add_argument("--prefill", type=int, default=128)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--layers", type=int, default=2)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-graphs", action="store_true")
ap.add_argument("--debug-lib", action="store_true")
a = ap.parse_args()
if a.debug_lib:
    _lib.use_debug_library(True)
base = named_dims("block_main_b4_1.2b")
dims = dataclasses.replace(base, name="diag", bd=dataclasses.replace(base.bd, layers=a.layers), td=dataclasses.replace(base.td, layers=a.layers))
torch.manual_seed(0)
with torch.device("cuda"):
    model = load_block_transformer_from_dims(dims, "bf16", keep_unused_embed_in=False)
model.token_decoder.generation_config.eos_token_id = None
model.use_graphs = not a.no_graphs
diag = torch.zeros(8 + 64 * 8, dtype=torch.int32, device="cuda")
if a.debug_lib:
    _lib.lib().bt_debug_sk_diag_buffer(diag.data_ptr())
ids = torch.randint(0, dims.vocab, (a.batch, a.prefill), generator=torch.Generator().manual_seed(1)).cuda()
MAGIC = 0x534b544f


def scan(eng):
    G = torch.cuda.get_device_properties(0).multi_processor_count
    out = []
    for name, t in (("ws_tok", eng.tok.ws), ("ws_step", eng.ws_step), ("ws_prefill", getattr(eng, "ws_prefill", None))):
        if t is None:
            continue
        w = t.view(torch.int32) if t.numel() % 4 == 0 else t[: t.numel() // 4 * 4].view(torch.int32)
        hit = (w == (MAGIC if MAGIC < 2 ** 31 else MAGIC - 2 ** 32)).nonzero().flatten()
        for h in hit[:4].tolist():
            out.append((name, h, [int(v) & 0xffffffff for v in w[h:h + 4].tolist()]))
    return out


for rep in range(a.reps):
    t0 = time.perf_counter()
    try:
        out = model.generate(ids, max_length=a.prefill + a.decode)
        torch.cuda.synchronize()
        print(f"rep {rep}: ok {time.perf_counter() - t0:.2f}s", flush=True)
    except _lib.BtError as e:
        print(f"rep {rep}: ERROR after {time.perf_counter() - t0:.2f}s: {str(e)[:80]}", flush=True)
        n = int(diag[0].item())
        print(f"   device diag: {n} time-outs recorded", flush=True)
        for r in diag[8:8 + 8 * min(n, 16)].view(-1, 8).tolist():
            print("      vw %d wave %d mode %d items/wg %d M %d N %d nk %d C %d" % tuple(r), flush=True)
        diag.zero_()
        for name, h, words in scan(model._engine):
            vw, w2, w3 = words[1], words[2], words[3]
            print(f"   sticky in {name} at word {h}: vw={vw} wave={w2 & 0xff} mode={(w2 >> 8) & 0xff} items_per_wg={w2 >> 16} tiles_m={w3 & 0xffff} tiles_n={w3 >> 16}", flush=True)
