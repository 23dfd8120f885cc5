"""The two-level decode engine behind ``BlockTransformer.generate`` (reference model/block_transformer.py:279-449).

State lives in preallocated HBM buffers with fixed addresses, described once to the C library by a ``bt_engine_desc``
(include/bt_hip.h); one *block step* is then a fixed launch sequence issued by C entry points:

    bt_token_block_enqueue:  expand ctx -> token decoder step 0 (q = E) -> logits -> argmax/sample -> embed_in ->
                             ... L tokens -> finish_block (EOS/pad rules, write ids, append block-mask column)
    bt_roll_over_enqueue:    embedder lookup+concat of the new block -> block decoder step (RoPE + append global KV at
                             row n, attention over n+1 block positions) -> ctx ; n += 1

The Python side only captures those calls into hipGraphs once and replays them per block; everything that changes between
replays (the number of cached block positions n, finished flags, the sampler's draw counter) is a device scalar read by
the kernels.  Finished sequences are masked inside an engine (fixed batch => static shapes => graph replay); the reference
instead compacts the batch with boolean indexing every block (model/block_transformer.py:381-398,421; C13/C14).  Compaction
here happens at bucket granularity: with EOS handling on, the per-block host look at ``finished[]`` (the reference's own sync,
:415) counts the live sequences, and once at most half of an engine's rows are live the loop continues on a smaller engine
(``DecodeEngine.compacted``).  Since round 4 that is ONE C call, ``bt_compact_enqueue``: the live rows' global KV moves to the
front of the SAME cache storage in place (no temporary), and the small engine is a *view* of the root engine -- the same
activation buffers, workspaces and local KV, its own graphs and a few KB of per-row state -- that writes its blocks straight
into the root's ``out_ids`` through ``bt_engine_desc.row_map`` (original row of each compacted row).  The sampler keys its RNG
streams by the same map, so compaction also runs under ``do_sample`` and with a streamer (the path inference_demo.py:62-83
takes); the streamer then receives the live rows only, like the reference's token decoder, which only ever sees
``decode_mask`` rows.  Local-KV "reset" costs nothing: every token block starts writing the local cache at row 0 again.

The streamer gets the reference's ``decode_mask`` rows of every block (the sequences unfinished when the block started), with or
without compaction, and their original row indices (``token_block_streamed``).

Measured-negative mechanisms are NOT in this module: the half-batch two-stream ``PipelinedEngine`` lives in
``tools/experiments/pipelined_engine.py`` (tests and A/B only; plugged in through ``BlockTransformer.engine_factory``), the
second stream for the parallel-residual branch exists only in the debug build of the library (``libbt_hip_dbg.so``).
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import List, Optional

import torch

from . import _lib, ops
from ._lib import bt_engine_desc, check, current_stream, dtype_code, lib, ptr
from .runtime import DecoderRuntime, Graph, KVCache

DEFAULT_ROPE_POLICY = "hf439_generate"


def td_position_table(policy: str, n_exp: int, block_length: int):
    """RoPE positions of the adapted token-decoder sequence [BE_1..BE_E, t_0..t_{L-2}].

    ``hf439_generate`` (default of generate()) = what the reference's *cached* generation computes: transformers 4.39.3
    ``GPTNeoXForCausalLM.prepare_inputs_for_generation`` derives ``position_ids = cumsum(attention_mask) - 1`` from the
    un-adapted mask and the reference forwards it through **kwargs into ``gpt_neox`` (model/token_decoder/gpt_neo_x.py:76-90
    -> base.py:158,301-303 -> gpt_neo_x.py:59-71): the E prefix rows all rotate at position 0 (a [B,1] position id
    broadcast over the q = E rows), token t at t+1.  Pinned by golden G9 (the reference's own classes evaluated at
    exactly these position ids).
    ``train`` = HF default arange over the adapted sequence: what a direct (no ``position_ids``) forward — teacher forcing,
    ``BlockTransformer.forward`` — computes; pinned by goldens G5/G6/G7.
    """
    if policy == "train":
        return list(range(n_exp + block_length - 1))
    if policy == "hf439_generate":
        return [0] * n_exp + list(range(1, block_length))
    raise ValueError(f"unknown RoPE position policy {policy!r}")


def _gen_fields(gen, cfg_pad: int):
    """(gen_eos, gen_pad, do_sample, temperature, top_k, seed) from a GenerationConfig-like object."""
    gen_eos = -1 if gen is None or gen.eos_token_id is None else int(gen.eos_token_id)
    gen_pad = cfg_pad
    if gen is not None and gen.pad_token_id is not None:
        gen_pad = int(gen.pad_token_id)
    elif gen_eos >= 0:
        gen_pad = gen_eos
    do_sample = bool(gen is not None and gen.do_sample)
    temperature = float((gen.temperature if gen is not None else None) or 1.0)
    top_k = int((gen.top_k if gen is not None else None) or 0)
    seed = int((getattr(gen, "seed", 0) if gen is not None else 0) or 0)
    return gen_eos, gen_pad, do_sample, temperature, top_k, seed


class TokenBlockState:
    """Token-decoder side of a block step: buffers + the token half of the descriptor.  Also serves
    ``GPTNeoXTokenDecoder.generate`` (reference model/token_decoder/base.py:399-453) on its own."""

    def __init__(self, token_decoder, batch: int, device, rope_policy: str = DEFAULT_ROPE_POLICY):
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
        table = td_position_table(rope_policy, self.E, self.L)
        self.pos_table = torch.tensor(table, dtype=torch.int32, device=device)
        self.rope = ops.rope_tables(max(table) + 1, self.rt.dims.rot, float(getattr(cfg, "rotary_emb_base", 10000.0)), device)
        # step 0 runs B*E prefix rows, steps 1.. run B rows (which may want split-K scratch when B <= 256): size for both
        self.ws = torch.empty(max(self.rt.workspace_bytes(batch * self.E), self.rt.workspace_bytes(batch)), dtype=torch.uint8,
                              device=device)
        self.sample_step = torch.zeros(1, dtype=torch.int32, device=device)
        # split-K scratch for the skinny expansion GEMM (small batches stream the weights with every CU)
        self.scratch = (torch.empty(16 * batch * self.E * self.H, dtype=torch.float32, device=device)
                        if batch <= 256 else None)
        self._desc: Optional[bt_engine_desc] = None   # stand-alone descriptor (token_decoder.generate)

    @classmethod
    def view(cls, parent: "TokenBlockState", batch: int) -> "TokenBlockState":
        """The first ``batch`` sequences of ``parent``, sharing its buffers (a compacted engine): nothing on the token side
        carries state across a roll-over except the block's raw tokens, which the view owns."""
        assert 0 < batch <= parent.B
        self = cls.__new__(cls)
        self.__dict__.update(parent.__dict__)
        self.B = batch
        self.local = KVCache.rows_of(parent.local, batch)
        dev = parent.raw.device
        self.raw = torch.zeros(batch, parent.L, dtype=torch.int64, device=dev)
        need = max(self.rt.workspace_bytes(batch * self.E), self.rt.workspace_bytes(batch))
        if parent.ws.numel() < need:      # a <= 256-row view of a large engine: its own split-K scratch (mid-batch path)
            self.ws = torch.empty(need, dtype=torch.uint8, device=dev)
        if batch <= 256 and parent.scratch is None:
            self.scratch = torch.empty(16 * batch * self.E * self.H, dtype=torch.float32, device=dev)
        self._desc = None
        return self

    def weight_key(self):
        td = self.td
        return (self.rt.weight_key(), td.expansion_layer.weight.data_ptr(),
                0 if td.expansion_layer.bias is None else td.expansion_layer.bias.data_ptr(),
                td.gpt_neox.embed_in.weight.data_ptr(), td.embed_out.weight.data_ptr())

    def fill_desc(self, d: bt_engine_desc) -> None:
        td = self.td
        for p in (td.expansion_layer.weight, td.gpt_neox.embed_in.weight, td.embed_out.weight):
            if p.dtype != self.dtype or not p.is_contiguous() or not p.is_cuda:
                raise _lib.BtError("token-decoder parameters must be contiguous, on the GPU and share the stack's dtype")
        d.dtype = dtype_code(self.dtype)
        d.L, d.E, d.vocab, d.B = self.L, self.E, self.V, self.B
        d.td_dims = self.rt.dims
        d.td_w = self.rt.weights()
        d.exp_w = td.expansion_layer.weight.data_ptr()
        d.exp_b = ptr(td.expansion_layer.bias)
        d.embed_in = td.gpt_neox.embed_in.weight.data_ptr()
        d.embed_out = td.embed_out.weight.data_ptr()
        d.lk, d.lv = self.local.k.data_ptr(), self.local.v.data_ptr()
        d.l_layer_stride, d.l_nmax = self.local.layer_stride, self.local.n_max
        d.l_cos, d.l_sin = self.rope[0].data_ptr(), self.rope[1].data_ptr()
        d.l_pos_table = self.pos_table.data_ptr()
        d.x_pre, d.x_tok, d.hid, d.logits = (self.x_pre.data_ptr(), self.x_tok.data_ptr(), self.hid.data_ptr(),
                                             self.logits.data_ptr())
        d.raw = self.raw.data_ptr()
        d.ws_tok, d.ws_tok_bytes = self.ws.data_ptr(), self.ws.numel()
        d.exp_scratch = ptr(self.scratch)
        d.exp_scratch_bytes = 0 if self.scratch is None else self.scratch.numel() * 4
        d.sample_step = self.sample_step.data_ptr()

    @staticmethod
    def set_gen(d: bt_engine_desc, gen, cfg_pad: int):
        gen_eos, gen_pad, do_sample, temperature, top_k, seed = _gen_fields(gen, cfg_pad)
        d.do_sample, d.temperature, d.top_k, d.seed = int(do_sample), temperature, top_k, seed
        d.config_pad, d.gen_eos, d.gen_pad = int(cfg_pad), gen_eos, int(gen_pad)
        return gen_eos, gen_pad, do_sample, temperature, top_k, seed

    def run(self, ctx: torch.Tensor, count: int, gen=None, on_token=None):
        """ctx [B, H_bd] -> raw[:, :count] token ids (stand-alone use: no finish step, no global state)."""
        key = self.weight_key()
        if self._desc is None or self._desc_key != key:
            self._desc = bt_engine_desc()
            self.fill_desc(self._desc)
            self._desc_key = key
        d = self._desc
        if ctx.dtype != self.dtype or not ctx.is_contiguous():
            raise _lib.BtError(f"block embeddings must be contiguous {self.dtype}")
        d.ctx = ptr(ctx)
        d.bd_dims.hidden = ctx.shape[-1]
        self.set_gen(d, gen, int(self.td.config.pad_token_id or 0))
        if on_token is None:
            check(lib().bt_token_block_enqueue(ctypes.byref(d), 0, count, 0, current_stream()), "bt_token_block_enqueue")
        else:
            for t in range(count):
                check(lib().bt_token_block_enqueue(ctypes.byref(d), t, t + 1, 0, current_stream()), "bt_token_block_enqueue")
                on_token(self.raw[:, t])
        return self.raw


class DecodeEngine:
    """Everything ``BlockTransformer.generate`` needs for one (batch, max_blocks) problem size on ONE stream."""

    pipelined = False

    # compaction (EOS handling on; greedy, sampling and streaming alike since round 4): continue on a smaller engine once <= half of the rows are live, down to this batch
    MIN_COMPACT_BATCH = 64
    COMPACT_BUCKET = 32           # compacted batches are multiples of this (bounds the number of engine shapes / graph sets)

    def __init__(self, model, batch: int, max_blocks: int, rope_policy: str = DEFAULT_ROPE_POLICY,
                 prefill_rows: Optional[int] = None, stream: Optional[torch.cuda.Stream] = None,
                 view_of: Optional["DecodeEngine"] = None):
        self.model = model
        if view_of is not None:
            self._init_view(view_of, batch)
            return
        bd, td = model.block_decoder, model.token_decoder
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
        self._root = None         # views hold a weak reference to their root (no reference cycle: an engine is 100s of GB)
        self._children = {}       # compacted views by batch size (state buffers and graphs are reused across generate() calls)
        self.row_map: Optional[torch.Tensor] = None     # identity
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=dev)   # sticky device-side guard bits (BT_DEV_ERR_*)
        # second stream + fork / join events for the parallel-residual branch of large stacks (bt_attn_args.aux_stream): measured
        # -0.2 ... -0.3 % end to end in round 4 -> the product library ignores them; only the debug build (tests / A-B) forks
        self.aux = None
        if getattr(model, "two_branch", None) and _lib._debug_active:
            aux = torch.cuda.Stream(device=dev)
            evs = [torch.cuda.Event(), torch.cuda.Event()]
            for ev in evs:
                ev.record(aux)             # torch creates the hipEvent lazily, on the first record
            self.aux = (aux, evs[0], evs[1])
        self.rope_policy = rope_policy
        self.tok = TokenBlockState(td, batch, dev, rope_policy)
        self.rope = ops.rope_tables(self.n_max, self.bd_rt.dims.rot,
                                    float(getattr(bd.config, "rotary_emb_base", 10000.0)), dev)
        self.x_blk = torch.empty(batch, self.H, dtype=dt, device=dev)    # embedded new block (block decoder input)
        self.ctx = torch.empty(batch, self.H, dtype=dt, device=dev)      # block decoder output = context embedding
        self.out_ids = torch.zeros(batch, max_blocks, self.L, dtype=torch.int64, device=dev)
        self.mask = torch.ones(batch, self.n_max + 1, dtype=torch.uint8, device=dev)
        self.finished = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.finished_prev = torch.zeros(batch, dtype=torch.int32, device=dev)
        # graphs cannot be captured on the null stream
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        self.n_cur = torch.zeros(1, dtype=torch.int32, device=dev)       # cached block positions == index of next block
        # prefill is chunked over sequences so the activation workspace stays bounded
        self.prefill_rows = int(prefill_rows or os.environ.get("BT_PREFILL_ROWS", 32768))
        self.ws_step = torch.empty(self.bd_rt.workspace_bytes(batch), dtype=torch.uint8, device=dev)
        self.ws_prefill: Optional[torch.Tensor] = None
        self.x_prefill: Optional[torch.Tensor] = None
        # split-N ("flash-decoding") for the global step when B*heads is too small to fill 256 CUs; the library raises the
        # factor on its own once a split's score row would not fit in LDS (very long contexts) -> same rule, same size here
        hint = 1
        work = batch * heads
        if work < 512 and self.n_max >= 256:
            hint = int(min(16, max(1, 1024 // max(work, 1)), max(1, self.n_max // 64)))
        self.n_split = int(lib().bt_attention_split_count(self.H // heads, self.n_max, hint))
        self.split_ws = None
        if self.n_split > 1:
            self.split_ws = torch.empty(batch * heads * self.n_split * (self.H // heads + 2), dtype=torch.float32,
                                        device=dev)
        self.desc = bt_engine_desc()
        self._desc_key = None
        self.graphs = {}
        self._graph_key = None

    def _init_view(self, root: "DecodeEngine", batch: int) -> None:
        """A compacted engine: the first ``batch`` sequence slots of ``root`` (same KV storage, activation buffers, workspaces,
        stream, out_ids) with its own per-row state (mask, finished flags, raw tokens, n_cur, row_map) and its own graphs."""
        assert root.root is root and 0 < batch <= root.B
        self._root = weakref.ref(root)
        self.bd_rt, self.device, self.dtype = root.bd_rt, root.device, root.dtype
        self.B, self.NB, self.L, self.H, self.n_max = batch, root.NB, root.L, root.H, root.n_max
        dev = self.device
        self.glob = KVCache.rows_of(root.glob, batch)
        self._children = None
        self.rope_policy = root.rope_policy
        self.tok = TokenBlockState.view(root.tok, batch)
        self.rope = root.rope
        self.x_blk, self.ctx = root.x_blk[:batch], root.ctx[:batch]
        self.out_ids = root.out_ids                       # written at row_map[b]
        self.mask = torch.ones(batch, self.n_max + 1, dtype=torch.uint8, device=dev)
        self.finished = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.finished_prev = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.row_map = torch.full((batch,), -1, dtype=torch.int64, device=dev)
        self.err_flag = root.err_flag
        self.aux = root.aux
        self.stream = root.stream
        self.n_cur = torch.zeros(1, dtype=torch.int32, device=dev)
        self.prefill_rows = root.prefill_rows
        need = self.bd_rt.workspace_bytes(batch)
        self.ws_step = root.ws_step if root.ws_step.numel() >= need else torch.empty(need, dtype=torch.uint8, device=dev)
        self.ws_prefill = self.x_prefill = None
        heads = self.glob.k.shape[2]
        hint, work = 1, batch * heads
        if work < 512 and self.n_max >= 256:
            hint = int(min(16, max(1, 1024 // max(work, 1)), max(1, self.n_max // 64)))
        self.n_split = int(lib().bt_attention_split_count(self.H // heads, self.n_max, hint))
        self.split_ws = None
        if self.n_split > 1:
            self.split_ws = torch.empty(batch * heads * self.n_split * (self.H // heads + 2), dtype=torch.float32, device=dev)
        self.desc = bt_engine_desc()
        self._desc_key = None
        self.graphs = {}
        self._graph_key = None

    @property
    def root(self) -> "DecodeEngine":
        return self if self._root is None else self._root()

    # ---- descriptor ----------------------------------------------------------------------------------------------------------
    def weight_key(self):
        emb = self.model.embedder.embeddings.weight
        return (self.bd_rt.weight_key(), self.tok.weight_key(), emb.data_ptr())

    def refresh_desc(self) -> None:
        """(Re)fill the C descriptor when parameters moved (``.to()``, dtype cast, checkpoint load with assign=True)."""
        key = self.weight_key()
        if key == self._desc_key:
            return
        d = self.desc
        self.tok.fill_desc(d)
        emb = self.model.embedder.embeddings.weight
        if emb.dtype != self.dtype or not emb.is_contiguous() or not emb.is_cuda:
            raise _lib.BtError("embedder table must be contiguous, on the GPU and share the decoders' dtype")
        d.bd_dims = self.bd_rt.dims
        d.bd_w = self.bd_rt.weights()
        d.emb_table, d.emb_width = emb.data_ptr(), emb.shape[1]
        d.gk, d.gv = self.glob.k.data_ptr(), self.glob.v.data_ptr()
        d.g_layer_stride, d.g_nmax = self.glob.layer_stride, self.glob.n_max
        d.g_cos, d.g_sin = self.rope[0].data_ptr(), self.rope[1].data_ptr()
        d.x_blk, d.ctx = self.x_blk.data_ptr(), self.ctx.data_ptr()
        d.out_ids, d.nb_max = self.out_ids.data_ptr(), self.NB
        d.n_cur, d.finished, d.finished_prev = self.n_cur.data_ptr(), self.finished.data_ptr(), self.finished_prev.data_ptr()
        d.mask_ld = self.mask.stride(0)
        d.ws_step, d.ws_step_bytes = self.ws_step.data_ptr(), self.ws_step.numel()
        d.n_split = self.n_split
        d.split_ws = ptr(self.split_ws)
        d.split_ws_bytes = 0 if self.split_ws is None else self.split_ws.numel() * 4
        d.row_map = ptr(self.row_map)
        d.err_flag = self.err_flag.data_ptr()
        if self.aux is not None:
            d.aux_stream, d.ev_fork, d.ev_join = self.aux[0].cuda_stream, self.aux[1].cuda_event, self.aux[2].cuda_event
        self._desc_key = key
        self.graphs = {}          # captured graphs bake in the old parameter pointers

    def configure(self, gen, cfg_pad: int, use_mask: bool):
        """Per-generate() settings: sampler / EOS fields of the descriptor; invalidates the graphs when they change (the
        captured launches hold them by value)."""
        self.refresh_desc()
        fields = TokenBlockState.set_gen(self.desc, gen, cfg_pad)
        self.desc.mask = self.mask.data_ptr() if use_mask else None
        # greedy steps hand the row maxima straight from the logits GEMM's epilogue to the argmax reduction; the [B, V] logits
        # matrix is only written when the caller asked to keep it (BlockTransformer.keep_logits, HF's output_scores analogue)
        skip = not bool(getattr(self.model, "keep_logits", False))
        self.desc.skip_logits = int(skip)
        key = (use_mask, int(cfg_pad), skip) + fields
        if key != self._graph_key:
            self.graphs = {}
            self._graph_key = key
        return fields[0]   # gen_eos

    # ---- phases ------------------------------------------------------------------------------------------------------------
    def prefill(self, block_ids: torch.Tensor, block_mask: torch.Tensor, scratch=None):
        """Embed all prompt blocks and run the block decoder over them (causal, q_len = nb); ctx <- last hidden row.
        ``scratch`` = (x, ws) tensors shared between the shards of a pipelined engine."""
        B, nb, L = block_ids.shape
        self.out_ids[:, :nb].copy_(block_ids)
        self.mask.fill_(1)
        self.mask[:, :nb].copy_(block_mask.to(torch.uint8))
        self.finished.zero_()
        self.finished_prev.zero_()
        self.n_cur.fill_(nb)
        self.tok.sample_step.zero_()    # the sampler's draw counter restarts: a generate() call is a function of (inputs, seed)
        bc = max(1, min(B, self.prefill_rows // max(nb, 1)))
        if scratch is None:
            need = self.bd_rt.workspace_bytes(bc * nb)
            if self.ws_prefill is None or self.ws_prefill.numel() < need:
                self.ws_prefill = None
                self.ws_prefill = torch.empty(need, dtype=torch.uint8, device=self.device)
            if self.x_prefill is None or self.x_prefill.numel() < bc * nb * self.H:
                self.x_prefill = None
                self.x_prefill = torch.empty(bc * nb, self.H, dtype=self.dtype, device=self.device)
            scratch = (self.x_prefill, self.ws_prefill)
        x, ws = scratch
        ids = block_ids.contiguous()
        for b0 in range(0, B, bc):
            n = min(B, b0 + bc) - b0
            check(lib().bt_prefill_enqueue(ctypes.byref(self.desc), ptr(ids), nb, b0, n, ptr(x), ptr(ws), ws.numel(),
                                           current_stream()), "bt_prefill_enqueue")

    def prefill_scratch_shape(self, nb: int):
        bc = max(1, min(self.B, self.prefill_rows // max(nb, 1)))
        return bc * nb, self.bd_rt.workspace_bytes(bc * nb)

    def _enqueue(self, what: str, count: int = 0):
        d = ctypes.byref(self.desc)
        if what == "tok":
            check(lib().bt_token_block_enqueue(d, 0, count, count, current_stream()), "bt_token_block_enqueue")
        elif what == "roll":
            check(lib().bt_roll_over_enqueue(d, current_stream()), "bt_roll_over_enqueue")
        else:   # "step": a full token block and the roll-over behind it
            check(lib().bt_block_step_enqueue(d, count, 1, current_stream()), "bt_block_step_enqueue")

    def launch(self, what: str, count: int, use_graphs: bool):
        """Run one unit ("tok" / "roll" / "step") on torch's current stream, through its captured graph when allowed."""
        if not use_graphs or (what != "roll" and count != self.L):
            return self._enqueue(what, count)
        g = self.graphs.get(what)
        if g is None:
            with Graph() as g:
                self._enqueue(what, count)
            self.graphs[what] = g
        g.launch()

    def token_block_streamed(self, count: int, streamer, n_real: Optional[int] = None, eos_on: bool = False):
        """Eager token block with one host callback per token (HF streamer protocol, inference_demo.py:69-83).  The streamer gets
        exactly the rows the reference hands it in EVERY block -- the ``decode_mask`` rows, i.e. the sequences that were unfinished
        when the block started (model/block_transformer.py:389-398) -- whether or not the engine has been compacted (one
        protocol before and after a compaction), in ascending ORIGINAL row order; ``block_start`` also receives those original row
        indices (``row_map`` of a compacted engine) so a batch streamer can attribute rows.  ``n_real``: rows of a compacted engine
        that carry a sequence."""
        d = ctypes.byref(self.desc)
        n = self.B if n_real is None else n_real
        live = None                                             # None = every one of the n rows
        if eos_on:
            live = (self.finished[:n] == 0).nonzero().flatten()  # the reference's decode_mask of this iteration (a host sync per
            if int(live.numel()) == n:                           # block, which the EOS loop makes anyway)
                live = None
        rows = torch.arange(n, device=self.tok.raw.device) if live is None else live
        streamer.block_start(int(rows.numel()), rows if self.row_map is None else self.row_map[rows])
        for t in range(count):
            fin = count if t == count - 1 else 0
            check(lib().bt_token_block_enqueue(d, t, t + 1, fin, current_stream()), "bt_token_block_enqueue")
            streamer.put(self.tok.raw[:n, t] if live is None else self.tok.raw[live, t])
        streamer.block_end()

    # ---- the generate loop -------------------------------------------------------------------------------------------------
    def generate(self, block_ids: torch.Tensor, block_mask: torch.Tensor, max_length: int, added_pad: int, gen,
                 cfg_pad: int, use_graphs: bool = True, streamer=None, timers=None):
        """Runs prefill + block iterations; returns (out_ids[B, n_blocks, L], next_token_count of the last block)."""
        B, nb, L = block_ids.shape
        use_mask = (gen is not None and gen.eos_token_id is not None) or not bool(block_mask.all())
        gen_eos = self.configure(gen, cfg_pad, use_mask)
        per_iter = timers is not None and timers.per_iteration
        side = self.stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if timers is not None:
                timers.mark("prefill_start")
            self.prefill(block_ids, block_mask)
            if timers is not None:
                timers.mark("decode_start")
            n_blocks = nb
            # EOS handling needs a host look at finished[] between the token block and the roll-over (like the
            # reference, :415); without it a block step is ONE graph (token block + roll-over)
            fused_step = use_graphs and gen_eos < 0 and streamer is None and not per_iter
            compacting = gen_eos >= 0 and bool(getattr(self.model, "compact_finished", True))
            eng = self                    # the engine the loop currently runs on (self, or a compacted view of it)
            n_real = B                    # eng's rows that carry a sequence (the rest of a compacted engine is padding)
            self.compactions = []         # (block index, live sequences, new batch) of this call: diagnostics / tests
            while True:
                count = min(max_length - (n_blocks * L - added_pad), L)
                last = (n_blocks + 1) * L >= max_length
                if per_iter:
                    timers.mark_iter("token_decoder_start")
                if streamer is not None:
                    eng.token_block_streamed(count, streamer, n_real, eos_on=gen_eos >= 0)
                elif fused_step and not last and count == L:
                    eng.launch("step", L, True)
                    n_blocks += 1
                    continue
                else:
                    eng.launch("tok", count, use_graphs)
                if per_iter:
                    timers.mark_iter("token_decoder_end")
                n_blocks += 1
                if n_blocks * L >= max_length:
                    break
                if gen_eos >= 0:   # host sync per block, like the reference (:415)
                    live = (eng.finished[:n_real] == 0).nonzero().flatten()
                    n_live = int(live.numel())
                    if n_live == 0:
                        break
                    if compacting and eng.B >= self.MIN_COMPACT_BATCH and 2 * n_live <= eng.B:
                        # reference :381-398,421: only decode_mask rows go on
                        if eng is self:   # rows that drop out keep the pad id in every later block (:384)
                            self.out_ids[:, n_blocks:].fill_(int(cfg_pad))
                        child = self.compacted(eng, live, gen, cfg_pad)
                        self.compactions.append((n_blocks, n_live, child.B))
                        eng, n_real = child, n_live
                if per_iter:
                    timers.mark_iter("block_decoder_start")
                eng.launch("roll", 0, use_graphs and streamer is None)
                if per_iter:
                    timers.mark_iter("block_decoder_end")
            if timers is not None:
                timers.mark("decode_end")
        torch.cuda.current_stream().wait_stream(side)
        self.check_device_errors()
        return self.out_ids[:, :n_blocks], count

    def check_device_errors(self) -> None:
        """Read the sticky bits the device-side guards set (bt_engine_desc.err_flag).  The Python loop bounds the block count
        itself (n_blocks <= NB by construction), so this can only fire on an engine misuse -- it is the same check a C binder
        that replays its captured block step makes after its last replay."""
        bits = int(self.err_flag.item())
        if bits:
            with torch.inference_mode():
                self.err_flag.zero_()
            what = [n for b, n in ((_lib.BT_DEV_ERR_CACHE_FULL, "global KV cache full (roll-over past g_nmax)"),
                                   (_lib.BT_DEV_ERR_BLOCK_RANGE, "block index past out_ids / mask"),
                                   (_lib.BT_DEV_ERR_HANDOFF_TIMEOUT, "a stream-K GEMM hand-off timed out (the GPU is shared with another "
                                                                     "process or a profiler?): activations incomplete, tokens discarded"))
                    if bits & b]
            raise _lib.BtError("device-side guard tripped: " + ", ".join(what))

    # ---- compaction of finished sequences --------------------------------------------------------------------------------
    def compacted(self, eng: "DecodeEngine", live: torch.Tensor, gen, cfg_pad: int) -> "DecodeEngine":
        """The view of this (root) engine that continues only the sequences ``live`` (ascending row indices of ``eng`` = this
        engine or one of its views) after a token block: batch = live count rounded up to COMPACT_BUCKET (the surplus rows are
        inert padding: finished, attention skipped, row_map -1 -> nothing of theirs is written).  bt_compact_enqueue moves the
        live rows' global KV to the front of the shared cache storage in place and fills the view's per-row state."""
        assert self.root is self and eng.root is self
        n = int(live.numel())
        Bc = min(eng.B, -(-n // self.COMPACT_BUCKET) * self.COMPACT_BUCKET)
        child = self._children.get(Bc)
        if child is None:
            child = self._children[Bc] = DecodeEngine(self.model, Bc, self.NB, view_of=self)
        assert child is not eng
        child.configure(gen, cfg_pad, True)
        check(lib().bt_compact_enqueue(ctypes.byref(eng.desc), ctypes.byref(child.desc), ptr(live.contiguous()), n,
                                       current_stream()), "bt_compact_enqueue")
        return child
