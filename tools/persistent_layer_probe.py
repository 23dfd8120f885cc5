#!/usr/bin/env python
"""Round 6: should the <= 4-row decoder layer become ONE persistent launch with grid-wide hand-offs?

The <= 2-row layer is three dependent launches inside the block step's hipGraph ([QKV || h->4h GEMV] -> attention -> [both
residual-stream GEMVs]); each dependency is an ALL-TO-ALL edge (every workgroup of the next phase reads rows produced by all
workgroups of the previous one: the full q/k/v row, the full attention output + h4 row, the full residual row for the LayerNorm).
A persistent layer replaces each launch boundary by a grid-wide barrier.  This probe prices both on the box, at the layer's own
sizes, with the same per-phase work on both sides (every workgroup streams its slice of the phase's weight bytes with non-temporal
16-byte loads and publishes / re-reads a 128-byte record):

  launches:   a hipGraph of `rounds` phase kernels (256 workgroups x 256 threads), replayed; us per phase
  persistent: ONE launch, `rounds` phases separated by (a) one device-scope counter barrier, (b) an XCD-hierarchical barrier
              (CDNA guide recipe); us per phase

for phase sizes 0 (the bare edge), 4.5 MB (attention of a block-decoder step at n = 544), 42 MB (residual GEMVs), 58.7 MB (QKV + h->4h).
The difference (persistent - launches) per phase at equal streamed bytes is what a barrier costs MORE (or less) than a launch
boundary; the CDNA guide's price list bounds what run-ahead prefetch across the edge can give back (prefetch-credit: 1.2 - 5 us per
edge with a dedicated LDS-DMA loader wave, i.e. the full loader / consumer engine, measured 0.87 - 0.89x of launches on a
batch-1 Llama layer).  Output: one JSON line per (bytes, form).  Debug library only.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_transformer_amd import _lib  # noqa: E402
from block_transformer_amd._lib import check, current_stream, lib, ptr  # noqa: E402
from block_transformer_amd.runtime import Graph  # noqa: E402

_lib.use_debug_library(True)
DEV = "cuda"


def timed_us(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    G = torch.cuda.get_device_properties(0).multi_processor_count
    rounds = 96
    src = torch.randint(0, 2 ** 31 - 1, (64 * 1024 * 1024 // 4,), dtype=torch.int32, device=DEV)      # 64 MiB of "weights"
    sink = torch.zeros(G, dtype=torch.float32, device=DEV)
    ticks = torch.zeros(1, dtype=torch.int64, device=DEV)
    for phase_bytes in (0, 4_456_448, 41_943_040, 58_720_256):
        per_wg = (phase_bytes // G) // 16 * 16
        state = torch.zeros(64 * 9, dtype=torch.int32, device=DEV)
        records = torch.zeros(G * 32, dtype=torch.int32, device=DEV)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with Graph() as g:
                for r in range(rounds):
                    check(lib().bt_debug_phase_probe(ptr(records), ptr(src), per_wg, G, r, ptr(sink), current_stream()))
            us = timed_us(g.launch)
        rec = {"op": "persistent_layer_probe", "phase_bytes": per_wg * G, "workgroups": G, "rounds": rounds,
               "launches_us_per_phase": round(us / rounds, 3)}
        for mode, name in ((0, "one_counter"), (1, "xcd_hierarchical")):
            def run():
                state.zero_()
                check(lib().bt_debug_grid_barrier_probe(ptr(state), ptr(records), ptr(src), per_wg, G, rounds, mode, ptr(ticks), ptr(sink),
                                                        current_stream()))
            with torch.cuda.stream(s):
                us_p = timed_us(run)
                torch.cuda.synchronize()
                rec[f"persistent_{name}_us_per_phase"] = round(us_p / rounds, 3)
                rec[f"persistent_{name}_in_kernel_us_per_phase"] = round(int(ticks.item()) * 0.01 / rounds, 3)     # s_memrealtime: 100 MHz
        rec["streaming_floor_us_at_5p8_TBps"] = round(per_wg * G / 5.8e6, 3)
        rec["barrier_minus_boundary_us"] = round(rec["persistent_xcd_hierarchical_us_per_phase"] - rec["launches_us_per_phase"], 3)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
