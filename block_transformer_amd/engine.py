"""The two-level decode engine behind ``BlockTransformer.generate`` (reference model/block_transformer.py:279-449).

State lives in preallocated HBM buffers with fixed addresses so that one *block step* is a fixed launch sequence:

    graph A "token block":   expand ctx -> token decoder step 0 (q = E) -> logits -> argmax/sample -> embed_in ->
                             ... L tokens -> finish_block (EOS/pad rules, write ids, append block-mask column)
    graph B "roll-over":     embedder lookup+concat of the new block -> block decoder step (RoPE + append global KV at
                             row n, attention over n+1 block positions) -> ctx ; n += 1

Both are captured once into hipGraphs and replayed per block; everything that changes between replays (the number of
cached block positions n) is a device scalar read by the kernels.  Finished sequences are masked, not compacted
(reference compacts with boolean indexing + host syncs, C13/C14): fixed batch => static shapes => graph replay.

Local-KV "reset" costs nothing: every token block starts writing the local cache at row 0 again.
"""
from __future__ import annotations

import math
from typing import Optional

import os

import torch

from . import _lib, ops
from ._lib import check, current_stream, lib, ptr
from .runtime import DecoderRuntime, Graph, KVCache


def td_position_table(policy: str, n_exp: int, block_length: int):
    """RoPE positions of the adapted token-decoder sequence [BE_1..BE_E, t_0..t_{L-2}].

    ``train`` = HF default arange (the no-cache forward the model was trained with; golden-pinned).
    ``hf439_generate`` = what transformers 4.39.3 ``prepare_inputs_for_generation`` feeds through the reference's
    **kwargs during cached generation: prefix rows all at 0, token t at t+1 (unverifiable here).
    """
    if policy == "train":
        return list(range(n_exp + block_length - 1))
    if policy == "hf439_generate":
        return [0] * n_exp + list(range(1, block_length))
    raise ValueError(f"unknown RoPE position policy {policy!r}")


class TokenBlockState:
    """Buffers + launch sequence for decoding one block of <= L tokens from a context embedding."""

    def __init__(self, token_decoder, batch: int, device, rope_policy: str = "train"):
        td = token_decoder
        self.td = td
        self.B = batch
        self.policy = rope_policy
        self.rt: DecoderRuntime = td._rt
        cfg = td.config
        self.H = cfg.hidden_size
        self.E = td.n_expanded_emb
        self.L = td.block_length
        self.V = cfg.vocab_size
        dt = self.rt.dtype
        self.dtype = dt
        heads = cfg.num_attention_heads
        self.local = KVCache(cfg.num_hidden_layers, batch, heads, self.E + self.L - 1, self.H // heads, dt, device)
        self.x_pre = torch.empty(batch * self.E, self.H, dtype=dt, device=device)   # expanded prefix rows
        self.x_tok = torch.empty(batch, self.H, dtype=dt, device=device)            # embed_in(token)
        self.hid = torch.empty(batch, self.H, dtype=dt, device=device)
        self.logits = torch.empty(batch, self.V, dtype=dt, device=device)
        self.raw = torch.zeros(batch, self.L, dtype=torch.int64, device=device)
        self.pos_table = torch.tensor(td_position_table(rope_policy, self.E, self.L), dtype=torch.int32, device=device)
        self.rope = ops.rope_tables(max(self.pos_table.tolist()) + 1, self.rt.dims.rot,
                                    float(getattr(cfg, "rotary_emb_base", 10000.0)), device)
        # step 0 runs B*E prefix rows, steps 1.. run B rows (which may want split-K scratch when B <= 256): size for both
        self.ws = torch.empty(max(self.rt.workspace_bytes(batch * self.E), self.rt.workspace_bytes(batch)), dtype=torch.uint8,
                              device=device)
        self.sample_step = torch.zeros(1, dtype=torch.int32, device=device)
        # split-K scratch for the skinny expansion GEMM (small batches stream the weights with every CU)
        self.scratch = (torch.empty(16 * batch * self.E * self.H, dtype=torch.float32, device=device)
                        if batch <= 256 else None)

    def _pick(self, t: int, gen):
        if gen is not None and gen.do_sample:
            ops.topk_sample(self.logits, temperature=float(gen.temperature or 1.0), top_k=int(gen.top_k or 0),
                            seed=int(getattr(gen, "seed", 0) or 0), step=self.sample_step, out=self.raw[:, t])
            check(lib().bt_bump(ptr(self.sample_step), 1, current_stream()), "bt_bump")
        else:
            ops.argmax(self.logits, out=self.raw[:, t])

    def run(self, ctx: torch.Tensor, count: int, gen=None, on_token=None):
        """ctx [B, H_bd] -> raw[:, :count] token ids.  Pure launch sequence (capturable when on_token is None)."""
        td, rt = self.td, self.rt
        # expansion_layer: Conv1d(k=1) == Linear; [B, H_bd] -> [B, E*H] == E prefix rows per sequence
        ops.linear(ctx, td.expansion_layer.weight, td.expansion_layer.bias, out=self.x_pre.view(self.B, self.E * self.H),
                   scratch=self.scratch)
        rt.forward(self.x_pre, self.E, self.local, pos0=0, pos_table=self.pos_table, first_out_row=self.E - 1,
                   out=self.hid, ws=self.ws, rope=self.rope)
        for t in range(count):
            if t > 0:
                ops.embed_rows(self.raw[:, t - 1], td.gpt_neox.embed_in.weight, out=self.x_tok)
                rt.forward(self.x_tok, 1, self.local, pos0=self.E + t - 1, pos_table=self.pos_table, out=self.hid,
                           ws=self.ws, rope=self.rope)
            ops.linear(self.hid, td.embed_out.weight, None, out=self.logits)
            self._pick(t, gen)
            if on_token is not None:
                on_token(self.raw[:, t])
        return self.raw


class DecodeEngine:
    """Everything ``BlockTransformer.generate`` needs for one (batch, max_blocks) problem size."""

    def __init__(self, model, batch: int, max_blocks: int, rope_policy: str = "train", prefill_rows: Optional[int] = None):
        self.model = model
        bd, td, emb = model.block_decoder, model.token_decoder, model.embedder
        self.bd_rt: DecoderRuntime = bd._rt
        dev = self.bd_rt.device
        dt = self.bd_rt.dtype
        self.device, self.dtype = dev, dt
        self.B, self.NB = batch, max_blocks
        self.L = model.block_length
        self.H = bd.config.hidden_size
        heads = bd.config.num_attention_heads
        self.n_max = max(max_blocks, 1)
        self.glob = KVCache(bd.config.num_hidden_layers, batch, heads, self.n_max, self.H // heads, dt, dev)
        self.tok = TokenBlockState(td, batch, dev, rope_policy)
        self.rope = ops.rope_tables(self.n_max, self.bd_rt.dims.rot,
                                    float(getattr(bd.config, "rotary_emb_base", 10000.0)), dev)
        self.x_blk = torch.empty(batch, self.H, dtype=dt, device=dev)    # embedded new block (block decoder input)
        self.ctx = torch.empty(batch, self.H, dtype=dt, device=dev)      # block decoder output = context embedding
        self.out_ids = torch.zeros(batch, max_blocks, self.L, dtype=torch.int64, device=dev)
        self.mask = torch.ones(batch, self.n_max + 1, dtype=torch.uint8, device=dev)
        self.finished = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.finished_prev = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.stream = torch.cuda.Stream(device=dev)                       # graphs cannot be captured on the null stream
        self.n_cur = torch.zeros(1, dtype=torch.int32, device=dev)       # cached block positions == index of next block
        # prefill is chunked over sequences so the activation workspace stays bounded
        self.prefill_rows = int(prefill_rows or os.environ.get("BT_PREFILL_ROWS", 32768))
        self.ws_step = torch.empty(self.bd_rt.workspace_bytes(batch), dtype=torch.uint8, device=dev)
        self.ws_prefill: Optional[torch.Tensor] = None
        # split-N ("flash-decoding") for the global step when B*heads is too small to fill 256 CUs
        self.n_split = 1
        work = batch * heads
        if work < 512 and self.n_max >= 256:
            self.n_split = int(min(16, max(1, 1024 // max(work, 1)), max(1, self.n_max // 64)))
        self.split_ws = None
        if self.n_split > 1:
            self.split_ws = torch.empty(batch * heads * self.n_split * (self.H // heads + 2), dtype=torch.float32,
                                        device=dev)
        self.graph_tok: Optional[Graph] = None
        self.graph_roll: Optional[Graph] = None
        self._graph_key = None

    # ---- phases ------------------------------------------------------------------------------------------------------------
    def prefill(self, block_ids: torch.Tensor, block_mask: torch.Tensor, use_mask: bool):
        """Embed all prompt blocks and run the block decoder over them (causal, q_len = nb); ctx <- last hidden row."""
        B, nb, L = block_ids.shape
        self.out_ids[:, :nb].copy_(block_ids)
        self.mask.fill_(1)
        self.mask[:, :nb].copy_(block_mask.to(torch.uint8))
        self.finished.zero_()
        self.n_cur.fill_(nb)
        emb_w = self.model.embedder.embeddings.weight
        bc = max(1, min(B, self.prefill_rows // max(nb, 1)))
        need = self.bd_rt.workspace_bytes(bc * nb)
        if self.ws_prefill is None or self.ws_prefill.numel() < need:
            self.ws_prefill = None
            self.ws_prefill = torch.empty(need, dtype=torch.uint8, device=self.device)
        x = torch.empty(bc * nb, self.H, dtype=self.dtype, device=self.device)
        mode = _lib.ATTN_FLASH if (self.dtype == torch.bfloat16 and nb > 1) else _lib.ATTN_ROWS
        for b0 in range(0, B, bc):
            b1 = min(B, b0 + bc)
            n = b1 - b0
            xs = x[:n * nb]
            ops.embed_rows(block_ids[b0:b1].reshape(-1), emb_w, out=xs.view(n * nb * L, -1))
            self.bd_rt.forward(xs, nb, self.glob, b0=b0, pos0=0, mask=self.mask[b0:b1] if use_mask else None,
                               mode=mode, first_out_row=nb - 1, out=self.ctx[b0:b1], ws=self.ws_prefill,
                               rope=self.rope)

    def token_block(self, count: int, gen, cfg_pad: int, gen_eos: int, gen_pad: int, use_mask: bool, on_token=None):
        """Decode <= L tokens from ctx, then finish the block on device (EOS/pad rules, ids, finished flags, mask)."""
        self.tok.run(self.ctx, count, gen, on_token)
        check(lib().bt_finish_block(self.B, self.L, count, ptr(self.tok.raw), ptr(self.out_ids), self.NB,
                                    ptr(self.n_cur), ptr(self.finished), ptr(self.finished_prev),
                                    ptr(self.mask) if use_mask else None, self.mask.stride(0), int(cfg_pad),
                                    int(gen_eos), int(gen_pad), current_stream()), "bt_finish_block")

    def roll_over(self, use_mask: bool):
        """Embed the block just finished (zero rows for sequences that were already finished before it,
        block_transformer.py:419-421), run the block-decoder step at cache row n, then n += 1."""
        ops.embed_rows(self.tok.raw.view(-1), self.model.embedder.embeddings.weight,
                       out=self.x_blk.view(self.B * self.L, -1), row_zero=self.finished_prev, rows_per_seq=self.L)
        self.bd_rt.forward(self.x_blk, 1, self.glob, pos0_dev=self.n_cur, mask=self.mask if use_mask else None,
                           mode=_lib.ATTN_ROWS, n_split=self.n_split, split_ws=self.split_ws, out=self.ctx,
                           ws=self.ws_step, rope=self.rope)
        check(lib().bt_bump(ptr(self.n_cur), 1, current_stream()), "bt_bump")

    # ---- the generate loop -------------------------------------------------------------------------------------------------
    def generate(self, block_ids: torch.Tensor, block_mask: torch.Tensor, max_length: int, added_pad: int, gen,
                 cfg_pad: int, use_graphs: bool = True, streamer=None, timers=None):
        """Runs prefill + block iterations; returns (out_ids[B, n_blocks, L], next_token_count of the last block)."""
        B, nb, L = block_ids.shape
        gen_eos = -1 if gen.eos_token_id is None else int(gen.eos_token_id)
        gen_pad = gen.pad_token_id if gen.pad_token_id is not None else (gen_eos if gen_eos >= 0 else cfg_pad)
        use_mask = gen_eos >= 0 or not bool(block_mask.all())
        on_token = None
        if streamer is not None:
            use_graphs = False
            on_token = streamer.put  # one [B] tensor per token, like HF generate's streamer
        key = (use_mask, gen_eos, gen_pad, cfg_pad, bool(gen.do_sample), float(gen.temperature or 1.0),
               int(gen.top_k or 0))
        if self._graph_key != key:
            self.graph_tok = self.graph_roll = None
            self._graph_key = key

        side = self.stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if timers is not None:
                timers.mark("prefill_start")
            self.prefill(block_ids, block_mask, use_mask)
            if timers is not None:
                timers.mark("decode_start")
            n_blocks = nb
            while True:
                count = min(max_length - (n_blocks * L - added_pad), L)
                if use_graphs and count == L:
                    if self.graph_tok is None:
                        with Graph() as g:
                            self.token_block(L, gen, cfg_pad, gen_eos, gen_pad, use_mask)
                        self.graph_tok = g
                    self.graph_tok.launch()
                else:
                    if streamer is not None:
                        streamer.block_start()
                    self.token_block(count, gen, cfg_pad, gen_eos, gen_pad, use_mask, on_token)
                    if streamer is not None:
                        streamer.block_end()
                n_blocks += 1
                if n_blocks * L >= max_length:
                    break
                if gen_eos >= 0 and bool(self.finished.all()):  # host sync per block, like the reference (:415)
                    break
                if use_graphs:
                    if self.graph_roll is None:
                        with Graph() as g:
                            self.roll_over(use_mask)
                        self.graph_roll = g
                    self.graph_roll.launch()
                else:
                    self.roll_over(use_mask)
            if timers is not None:
                timers.mark("decode_end")
        torch.cuda.current_stream().wait_stream(side)
        return self.out_ids[:, :n_blocks], count
