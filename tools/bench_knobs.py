#!/usr/bin/env python
"""Run bench.py with library debug knobs set first (A/B experiments), e.g.
    python tools/bench_knobs.py few_rows_max=8 gemv_rows=8 -- --batch 8 --steps 3 --no-cpu-baseline --no-roofline"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (libbt_hip must load after torch)
from block_transformer_amd import _lib  # noqa: E402
from block_transformer_amd._lib import lib  # noqa: E402

_lib.use_debug_library(True)   # the bt_debug_* knobs exist only in libbt_hip_dbg.so

sep = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
for kv in sys.argv[1:sep]:
    k, v = kv.split("=")
    getattr(lib(), "bt_debug_" + k)(int(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[sep + 1:]
if os.environ.get("BT_ENGINE") == "pipelined":   # experiment: two half-batch shards on two streams (tools/experiments/pipelined_engine.py)
    from block_transformer_amd.model.block_transformer import BlockTransformer
    from tools.experiments.pipelined_engine import PipelinedEngine
    _init = BlockTransformer.__init__

    def _patched(self, *a, **kw):
        _init(self, *a, **kw)
        self.engine_factory = PipelinedEngine
    BlockTransformer.__init__ = _patched
diag = None
if os.environ.get("BT_SK_DIAG") == "1":      # never-cleared record of stream-K hand-off time-outs (debug build), printed if the run dies
    diag = torch.zeros(8 + 64 * 8 + 64, dtype=torch.int32, device="cuda")
    lib().bt_debug_sk_diag_buffer(diag.data_ptr())
mid_tl = None
if os.environ.get("BT_MID_TL") == "1":       # timeline of the fused mid-batch QKV + h->4h launch (debug build): the LAST launch's stamps survive
    mid_tl = torch.zeros(512 * 8 * 16, dtype=torch.int64, device="cuda")
    lib().bt_debug_timeline_buffer(mid_tl.data_ptr())
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    if mid_tl is not None:
        import json
        torch.cuda.synchronize()
        raw = mid_tl.view(512, 8, 16).cpu()
        used = raw[:, 0, 0] != 0
        n_wg = int(used.sum())
        st = raw[used][:, :, :10].double()                       # [wg, wave, stamp]
        t0 = st[:, :, 0].min()
        rel = st[:, :, :9] - t0                                   # ticks since the first wave's entry (s_memtime: 100 MHz on this part)
        names = ["entry", "prologue_issued", "first_tile_opens", "quarter", "half", "three_quarters", "loop_done", "epilogue_starts", "epilogue_done"]
        rec = {"op": "mid_fused_timeline", "workgroups": n_wg, "unit": "s_memtime ticks (10 ns) since the first wave's entry"}
        for i, nme in enumerate(names):
            v = rel[:, :, i].flatten()
            rec[nme] = {"min": float(v.min()), "median": float(v.median()), "max": float(v.max())}
        for g, sl in (("group0", slice(0, 4)), ("group1", slice(4, 8))):
            d = st[:, sl]
            rec[g + "_median_ticks"] = {"entry_to_first_tile": float((d[:, :, 2] - d[:, :, 0]).median()),
                                         "k_loop": float((d[:, :, 6] - d[:, :, 2]).median()),
                                         "first_quarter": float((d[:, :, 3] - d[:, :, 2]).median()),
                                         "second_quarter": float((d[:, :, 4] - d[:, :, 3]).median()),
                                         "third_quarter": float((d[:, :, 5] - d[:, :, 4]).median()),
                                         "fourth_quarter": float((d[:, :, 6] - d[:, :, 5]).median()),
                                         "epilogue": float((d[:, :, 8] - d[:, :, 7]).median()),
                                         "whole": float((d[:, :, 8] - d[:, :, 0]).median())}
        rt = raw[used][:, 0, 9].double()
        rec["entry_spread_realtime_ticks"] = float(rt.max() - rt.min())
        print("[bench_knobs] " + json.dumps(rec), file=sys.stderr)
    if diag is not None:
        n = int(diag[0].item())
        print(f"[bench_knobs] stream-K time-outs recorded on the device: {n}", file=sys.stderr)
        for r in diag[8:8 + 8 * min(n, 16)].view(-1, 8).tolist():
            print("[bench_knobs]   vw %d wave %d mode %d items/wg %d M %d N %d nk %d C %d" % tuple(r), file=sys.stderr)
        m = int(diag[1].item())
        print(f"[bench_knobs] passes whose sticky word was non-zero at their end: {m}", file=sys.stderr)
        for r in diag[8 + 512:8 + 512 + 8 * min(m, 8)].view(-1, 8).tolist():
            addr = (r[6] & 0xffffffff) | ((r[7] & 0xffffffff) << 32)
            print("[bench_knobs]   pass rows %d q_len %d: sticky word at %#x holds %s, word before %s" % (r[0] & 0xffffff, (r[0] >> 24) & 0xff, addr,
                  [hex(v & 0xffffffff) for v in r[1:5]], hex(r[5] & 0xffffffff)), file=sys.stderr)
            import gc
            for obj in gc.get_objects():       # which live CUDA tensor (storage) contains that address?
                try:
                    if isinstance(obj, torch.Tensor) and obj.is_cuda:
                        st = obj.untyped_storage()
                        if st.data_ptr() <= addr < st.data_ptr() + st.nbytes():
                            print("[bench_knobs]      inside storage %#x + %d bytes (offset %d, %d bytes before its end), tensor shape %s dtype %s" % (
                                st.data_ptr(), st.nbytes(), addr - st.data_ptr(), st.data_ptr() + st.nbytes() - addr, tuple(obj.shape), obj.dtype), file=sys.stderr)
                except Exception:
                    pass
