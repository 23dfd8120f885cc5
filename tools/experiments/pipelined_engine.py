"""EXPERIMENT (measured negative, kept for A/B and tests only; not part of the product package -- VERDICT r4 #8).

``PipelinedEngine``: TWO half-batch ``DecodeEngine`` shards on two HIP streams, one token block apart, so that shard A's global-KV
stream (block-decoder step, HBM-bound, matrix cores idle) could overlap shard B's token-decoder GEMMs (MFMA-bound).  Measured on
MI355X (profiles/r2_bench_pipeline_{off,on,sync}.json, profiles/r2_overlap_probe_*.jsonl): 115.9k -> 113.3k
tok/s -- the attention grid and the persistent GEMM each fill the CUs, <= 30 % of the shorter phase hides even with
bounded-footprint kernels.  Plug it in with ``model.engine_factory = PipelinedEngine`` (``BlockTransformer._get_engine`` then
builds it for batches >= 2 without streamer / EOS handling); bit-identical tokens to the single-stream engine in fp32
(tests/test_model_gpu.py::test_half_batch_pipeline_*)."""
from __future__ import annotations

import os
import sys
from typing import List

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from block_transformer_amd import _lib  # noqa: E402
from block_transformer_amd.engine import DEFAULT_ROPE_POLICY, DecodeEngine  # noqa: E402


class PipelinedEngine:
    """Two half-batch ``DecodeEngine`` shards on two streams, offset by one token block, so that one shard's HBM-bound
    global-attention step overlaps the other's MFMA-bound token-decoder GEMMs.  Sequences are independent (SURVEY.md
    section 8e), so the result equals the single-engine result row for row."""

    pipelined = True

    def __init__(self, model, batch: int, max_blocks: int, rope_policy: str = DEFAULT_ROPE_POLICY, n_shards: int = 2):
        self.model = model
        self.B, self.NB = batch, max_blocks
        bounds = [(batch * i) // n_shards for i in range(n_shards + 1)]
        self.bounds = bounds
        self.shards: List[DecodeEngine] = [DecodeEngine(model, bounds[i + 1] - bounds[i], max_blocks, rope_policy)
                                           for i in range(n_shards)]
        for i, s in enumerate(self.shards):
            s.desc.sample_row0 = bounds[i]     # shards draw the RNG streams the whole batch would
        s0 = self.shards[0]
        self.device, self.dtype, self.L = s0.device, s0.dtype, s0.L
        self.tok = s0.tok             # policy / dims probe used by BlockTransformer._get_engine
        self.stream = s0.stream
        self.sync_phases = os.environ.get("BT_PIPELINE_SYNC", "0") == "1"
        self._scratch = None

    @property
    def ctx(self) -> torch.Tensor:
        return torch.cat([s.ctx for s in self.shards], dim=0)

    @property
    def finished(self) -> torch.Tensor:
        return torch.cat([s.finished for s in self.shards], dim=0)

    def _prefill_scratch(self, nb: int):
        rows, nbytes = max(s.prefill_scratch_shape(nb) for s in self.shards)
        sc = self._scratch
        if sc is None or sc[0].shape[0] < rows or sc[1].numel() < nbytes:
            self._scratch = None
            s0 = self.shards[0]
            self._scratch = (torch.empty(rows, s0.H, dtype=self.dtype, device=self.device),
                             torch.empty(nbytes, dtype=torch.uint8, device=self.device))
        return self._scratch

    def generate(self, block_ids: torch.Tensor, block_mask: torch.Tensor, max_length: int, added_pad: int, gen,
                 cfg_pad: int, use_graphs: bool = True, streamer=None, timers=None):
        if streamer is not None:
            raise _lib.BtError("streaming runs on the single-stream engine (BlockTransformer picks it when a streamer is given)")
        B, nb, L = block_ids.shape
        use_mask = (gen is not None and gen.eos_token_id is not None) or not bool(block_mask.all())
        gen_eos = -1
        for s in self.shards:
            gen_eos = s.configure(gen, cfg_pad, use_mask)
        if gen_eos >= 0:
            # HF stops a block once EVERY row has produced EOS -- a whole-batch condition the shards cannot evaluate
            # independently, and the per-block look at finished[] would have to be ordered behind every shard's stream
            raise _lib.BtError("EOS handling runs on the single-stream engine (BlockTransformer picks it when "
                               "generation_config.eos_token_id is set); PipelinedEngine needs eos_token_id = None")
        per_iter = timers is not None and timers.per_iteration
        main = self.stream
        main.wait_stream(torch.cuda.current_stream())
        scratch = self._prefill_scratch(nb)
        with torch.cuda.stream(main):
            if timers is not None:
                timers.mark("prefill_start")
            for i, s in enumerate(self.shards):   # prefill is MFMA-bound end to end: shards run back to back
                lo, hi = self.bounds[i], self.bounds[i + 1]
                s.prefill(block_ids[lo:hi], block_mask[lo:hi], scratch)
            if timers is not None:
                timers.mark("decode_start")
        # decode: shard i runs on its own stream, one token block behind shard i-1
        streams = [main] + [s.stream for s in self.shards[1:]]
        for st in streams[1:]:
            st.wait_stream(main)
        n_blocks = nb
        gate = None
        first = True
        while True:
            count = min(max_length - (n_blocks * L - added_pad), L)
            for i, s in enumerate(self.shards):
                with torch.cuda.stream(streams[i]):
                    # first block: every shard starts one token block behind the previous one, then they free-run (roll-over
                    # of one shard beside the token block of the other).  BT_PIPELINE_SYNC=1 keeps passing the baton for
                    # the whole decode: at most one shard is inside a token block (the MFMA-bound phase) at any time.
                    gated = first or self.sync_phases
                    if gated and gate is not None:
                        streams[i].wait_event(gate)
                    if per_iter and i == 0:
                        timers.mark_iter("token_decoder_start")
                    s.launch("tok", count, use_graphs)
                    if per_iter and i == 0:
                        timers.mark_iter("token_decoder_end")
                    if gated:
                        gate = torch.cuda.Event()
                        gate.record(streams[i])
            first = False
            n_blocks += 1
            if n_blocks * L >= max_length:
                break
            for i, s in enumerate(self.shards):
                with torch.cuda.stream(streams[i]):
                    if per_iter and i == 0:
                        timers.mark_iter("block_decoder_start")
                    s.launch("roll", 0, use_graphs)
                    if per_iter and i == 0:
                        timers.mark_iter("block_decoder_end")
        for st in streams[1:]:
            main.wait_stream(st)
        with torch.cuda.stream(main):
            if timers is not None:
                timers.mark("decode_end")
        torch.cuda.current_stream().wait_stream(main)
        out = torch.cat([s.out_ids[:, :n_blocks] for s in self.shards], dim=0)
        return out, count
