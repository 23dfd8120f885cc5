#!/usr/bin/env python
"""Round 6 probe: does a hipMemsetAsync node inside a replayed hipGraph zero ALL of its bytes?  (profiles/r6_handoff_false_positive.md)
Poison a buffer, memset `n` bytes at a 256-byte-aligned offset eagerly / through a captured graph replayed `reps` times with the buffer
re-poisoned in between (by a kernel inside the same graph, like a decoder pass rewriting the region), report the first non-zero byte."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_transformer_amd import _lib  # noqa: E402
from block_transformer_amd._lib import check, current_stream, lib, ptr  # noqa: E402
from block_transformer_amd.runtime import Graph  # noqa: E402

_lib.use_debug_library(True)
buf = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
off = 65536 * 4
for n in (4096, 8192, 8224, 8320, 12288 + 32):
    rec = {"op": "memset_graph_probe", "bytes": n}
    buf.fill_(0xAB)
    check(lib().bt_debug_memset(buf.data_ptr() + off, 0, n, current_stream()))
    torch.cuda.synchronize()
    nz = buf[off:off + n].nonzero()
    rec["eager_first_nonzero"] = int(nz[0]) if nz.numel() else None
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with Graph() as g:
            check(lib().bt_debug_memset(buf.data_ptr() + off, 0, n, current_stream()))
        bad = []
        for rep in range(50):
            buf.fill_(0xAB)
            g.launch()
            torch.cuda.synchronize()
            nz = buf[off:off + n].nonzero()
            if nz.numel():
                bad.append((rep, int(nz[0]), int(nz.numel())))
    rec["graph_replays_with_nonzero_bytes"] = len(bad)
    rec["first_bad"] = bad[:3]
    rec["bytes_behind_untouched"] = bool((buf[off + n:off + n + 64] == 0xAB).all())
    print(json.dumps(rec), flush=True)
