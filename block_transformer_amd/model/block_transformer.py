"""BlockTransformer — same constructor, attributes and generate() contract as the reference's
model/block_transformer.py:14-465, driving the MI355X decode engine (block_transformer_amd/engine.py).

Reference behaviours reproduced on purpose: Q2 (end test ignores initial padding), Q4 (lookup
embedder pads with id 0 and ignores masks), Q5/Q6 (finished rows: zero block embedding, mask column 0, pad tokens;
HF early stop inside a block), Q7 knobs live on token_decoder.config / .generation_config.  Q1 (the reference's
early-return path raises AttributeError through a mis-imported ``warning``) is implemented as intended: warn and
return the input unchanged."""
from __future__ import annotations

import warnings
from typing import Optional

import torch
from torch import nn

from ..engine import DEFAULT_ROPE_POLICY, DecodeEngine


class _Timers:
    """hipEvent phase marks for generate(benchmark=True) and bench.py (prefill vs decode split); with
    ``per_iteration`` also one start/end pair per block-decoder step and per token block, like the reference's
    ``block_decoder_starts/ends`` and ``token_decoder_starts/ends`` lists (model/block_transformer.py:347-353,428-444)."""

    def __init__(self, per_iteration: bool = False):
        self.events = {}
        self.per_iteration = per_iteration
        self.iter_events = {}

    def mark(self, name: str):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        self.events[name] = ev

    def mark_iter(self, name: str):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        self.iter_events.setdefault(name, []).append(ev)

    def ms(self, a: str, b: str) -> float:
        self.events[b].synchronize()
        return self.events[a].elapsed_time(self.events[b])

    def iter_ms(self, kind: str):
        starts, ends = self.iter_events.get(kind + "_start", []), self.iter_events.get(kind + "_end", [])
        out = []
        for a, b in zip(starts, ends):
            b.synchronize()
            out.append(a.elapsed_time(b))
        return out


class BlockTransformer(nn.Module):
    def __init__(self, embedder, block_decoder, token_decoder, token_mapper=None, use_token_decoding_loss=False,
                 use_block_decoding_loss=False, block_decoding_loss_weight=1.0, use_auto_encoding_loss=False,
                 auto_encoding_loss_weight=1.0, decoding_strategy=None):
        super().__init__()
        self.embedder = embedder
        self.token_decoder = token_decoder
        self.block_decoder = block_decoder
        if token_mapper is not None:
            raise NotImplementedError("token_mapper is None for every block_main_* config (same tokenizer on both "
                                      "sides); vocabulary mapping is outside the decode path")
        self.token_mapper = None
        self.block_length = self.token_decoder.block_length
        self.n_embedding_tokens = self.embedder.n_embedding_tokens
        self.projection_hidden_size = self.embedder.projection_hidden_size
        self.use_token_decoding_loss = use_token_decoding_loss
        if self.use_token_decoding_loss and self.token_decoder is None:
            raise ValueError("token_decoder must be specified when use_token_decoding_loss is True")
        self.use_block_decoding_loss = use_block_decoding_loss
        self.block_decoding_loss_weight = block_decoding_loss_weight
        self.use_auto_encoding_loss = use_auto_encoding_loss
        self.auto_encoding_loss_weight = auto_encoding_loss_weight
        self.decoding_strategy = decoding_strategy
        self.config = self.token_decoder.config  # for Evaluation (reference :48)
        self.use_graphs = True
        self.engine_factory = None    # experiments only (tools/experiments): an engine class used instead of DecodeEngine
        self.collect_timers = False  # bench.py: record phase events without the benchmark printout
        # greedy generate() keeps only token ids, like HF generate with output_scores=False: the logits GEMM reduces each row to
        # its argmax in its epilogue (bt_logits_argmax) and the [B, vocab] matrix is never written.  True = also materialise the
        # last step's logits in ``_engine.tok.logits`` (tests, debugging); the tokens are the same bit for bit either way.
        self.keep_logits = False
        # EOS handling: once at most half of the batch is still decoding, continue on a smaller engine with only the live
        # sequences (the reference compacts every block with decode_mask indexing, :381-398,421; engine.DecodeEngine.compacted).
        # False = finished rows stay in the batch, masked (same tokens; A/B and tests).
        self.compact_finished = True
        self._engine: Optional[DecodeEngine] = None
        self.last_timers: Optional[_Timers] = None

    # ---- input format helpers (pure integer tensor work; device-agnostic) ----------------------------------------------
    def preprocess_inputs_for_generation(self, input_ids: torch.LongTensor,
                                         attention_mask: Optional[torch.LongTensor] = None):
        """Vanilla -> block format (reference :227-266): left-pad to a multiple of block_length."""
        input_ids = input_ids.view(-1, input_ids.shape[-1])
        if attention_mask is None:
            attention_mask = input_ids.new_ones(input_ids.shape, dtype=torch.long)
            pad_id = self.embedder.config.pad_token_id
            if pad_id is not None:  # `input_ids == None` is a no-op in the reference (Q4)
                attention_mask[input_ids == pad_id] = 0
        else:
            attention_mask = attention_mask.view(-1, attention_mask.shape[-1])
        bs = input_ids.shape[0]
        rem = input_ids.shape[-1] % self.block_length
        pad_length = 0
        if rem != 0:
            pad_length = self.block_length - rem
            pad_id = self.embedder.config.pad_token_id
            input_ids = torch.nn.functional.pad(input_ids, (pad_length, 0), mode="constant",
                                                value=0 if pad_id is None else pad_id)
            attention_mask = torch.nn.functional.pad(attention_mask, (pad_length, 0), mode="constant", value=0)
        input_ids = input_ids.view(bs, -1, self.block_length)
        attention_mask = attention_mask.view(bs, -1, self.block_length)
        block_attention_mask = attention_mask.any(dim=-1).long()
        return {"input_ids": input_ids, "attention_mask": attention_mask,
                "block_attention_mask": block_attention_mask, "initial_block_padding": pad_length}

    @staticmethod
    def revert_block_input_ids_to_vanilla(input_ids, added_initial_block_padding, last_block_unfilled_length):
        """Block -> vanilla format (reference :268-277)."""
        input_ids = input_ids.reshape(-1, input_ids.shape[-2] * input_ids.shape[-1])
        input_ids = input_ids[:, added_initial_block_padding:]
        if last_block_unfilled_length > 0:
            input_ids = input_ids[:, :-last_block_unfilled_length]
        return input_ids

    def embed_blocks(self, input_ids, attention_mask):
        """[B, nb, L] ids -> [B, nb, n_embedding_tokens, projection_hidden_size] (reference :451-465)."""
        assert len(input_ids.shape) == 3 and len(attention_mask.shape) == 3
        batch_size, n_blocks, _ = input_ids.shape
        block_embeds = self.embedder(input_ids.reshape(-1, input_ids.shape[-1]),
                                     attention_mask.reshape(-1, attention_mask.shape[-1]))
        return block_embeds.view((batch_size, n_blocks) + block_embeds.shape[-2:])

    # ---- engine management ---------------------------------------------------------------------------------------------------
    def _get_engine(self, batch: int, max_blocks: int, streaming: bool = False):
        e = self._engine
        rt = self.block_decoder._rt
        policy = getattr(self.token_decoder, "rope_policy", DEFAULT_ROPE_POLICY)
        # single stream when a streamer needs per-token host callbacks, and when EOS handling is on: HF stops a block early
        # once EVERY row of the batch has produced EOS (Q6), a whole-batch condition two independent shards cannot evaluate
        # (``engine_factory``: experiments plug their own engine class in -- tools/experiments/pipelined_engine.py; None = the product)
        factory = self.engine_factory if (self.engine_factory is not None and not streaming and batch >= 2) else DecodeEngine
        if (e is None or e.B != batch or e.NB < max_blocks or e.dtype != rt.dtype or e.device != rt.device
                or e.tok.policy != policy or type(e) is not factory):
            self._engine = e = None  # free the old buffers before allocating the new ones
            e = self._engine = factory(self, batch, max_blocks, rope_policy=policy)
        return e

    def release_buffers(self):
        """Drop the engine (KV caches, workspaces, captured graphs) and return the blocks to the driver (a differently
        shaped engine -- other batch, single-stream vs half-batch shards -- needs contiguous room for its own caches)."""
        self._engine = None
        import gc
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()

    # ---- generate ----------------------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def generate(self, input_ids: torch.LongTensor, attention_mask: Optional[torch.LongTensor] = None,
                 block_attention_mask: Optional[torch.LongTensor] = None, max_length: int = 100,
                 benchmark: bool = False, streamer=None):
        """Inputs in vanilla format ([B, T]) or block format ([B, nb, L] with block_attention_mask).  Returns ids in the
        same format, exactly like the reference (:279-449)."""
        added_initial_block_padding = 0
        initial_shapes = {"input_ids": input_ids.shape,
                          "attention_mask": attention_mask.shape if attention_mask is not None else None,
                          "block_attention_mask": block_attention_mask.shape if block_attention_mask is not None else None}
        vanilla_mode = block_attention_mask is None
        if vanilla_mode:
            d = self.preprocess_inputs_for_generation(input_ids, attention_mask)
            input_ids, attention_mask = d["input_ids"], d["attention_mask"]
            block_attention_mask = d["block_attention_mask"]
            added_initial_block_padding = d["initial_block_padding"]
        else:
            if input_ids.shape[-1] != self.block_length:
                raise ValueError("input_ids.shape[-1] must be equal to self.block_length when block_attention_mask "
                                 "is given")
            if attention_mask is None:
                raise ValueError("attention_mask must be specified when block_attention_mask is specified")
        n_blocks, block_length = input_ids.shape[-2:]
        input_ids = input_ids.reshape(-1, n_blocks, block_length)
        attention_mask = attention_mask.reshape(-1, n_blocks, block_length)
        block_attention_mask = block_attention_mask.reshape(-1, n_blocks)
        batch_size = input_ids.shape[0]
        if input_ids.shape != attention_mask.shape:
            raise ValueError(f"input_ids and attention_mask must have equivalent shapes. Given: {initial_shapes}")
        if input_ids.shape[:2] != block_attention_mask.shape[-2:]:
            raise ValueError(f"input_ids and block_attention_mask must have equivalent shapes. Given: {initial_shapes}")
        if n_blocks * block_length - added_initial_block_padding > max_length:
            warnings.warn("max_length is less than the length of the input_ids. Returning original input_ids.")
            if vanilla_mode:
                input_ids = self.revert_block_input_ids_to_vanilla(input_ids, added_initial_block_padding, 0)
            return input_ids
        if self.n_embedding_tokens != 1:
            raise NotImplementedError("generation currently only implemented for n_embedding_tokens == 1")
        if not input_ids.is_cuda:
            raise RuntimeError("BlockTransformer.generate runs on the GPU only (libbt_hip.so has no CPU path): move the "
                               "inputs and the model to 'cuda'")

        # same loop arithmetic as the reference: one block per iteration until n_blocks * L >= max_length (Q2)
        total_blocks = n_blocks
        while True:
            total_blocks += 1
            if total_blocks * block_length >= max_length:
                break
        gen = self.token_decoder.generation_config
        engine = self._get_engine(batch_size, total_blocks, streaming=streamer is not None or gen.eos_token_id is not None)
        timers = _Timers(per_iteration=bool(benchmark)) if (benchmark or self.collect_timers) else None
        if benchmark:
            print(f"Benchmarking generation... (benchmark={benchmark})")
        if streamer is not None:
            streamer = _BlockStreamer(streamer, batch_size, self.token_decoder.config.eos_token_id)
        out, next_token_count = engine.generate(input_ids.contiguous(), block_attention_mask.contiguous(), max_length,
                                                added_initial_block_padding, gen,
                                                int(self.token_decoder.config.pad_token_id),
                                                use_graphs=self.use_graphs, streamer=streamer, timers=timers)
        out = out.clone()
        self.last_timers = timers
        if benchmark:
            # same report as the reference (:428-444): the first block-decoder time is the prefill, the others the
            # per-block global steps; one token-decoder time per generated block (L cached forwards each)
            bd = [timers.ms("prefill_start", "decode_start")] + timers.iter_ms("block_decoder")
            td = timers.iter_ms("token_decoder")
            print(" Generation time (ms) ".center(100, "-"))
            if getattr(engine, "pipelined", False):
                print("(two half-batch shards on two streams: per-iteration times are shard 0's and overlap shard 1's work)")
            print("Total time: ", timers.ms("prefill_start", "decode_end"))
            print("Block decoder times: ", bd)
            print("Token decoder times: ", td)
            print("Block decoder total time: ", sum(bd))
            print("Token decoder total time: ", sum(td))
        if vanilla_mode:
            out = self.revert_block_input_ids_to_vanilla(out, added_initial_block_padding,
                                                         last_block_unfilled_length=block_length - next_token_count)
        return out

    @torch.inference_mode()
    def forward(self, input_ids: torch.LongTensor, attention_mask: torch.LongTensor,
                block_attention_mask: torch.LongTensor, labels=None, skip_padding_blocks: bool = True):
        """Scoring path (reference :50-225, inference use: lm_eval wrappers and util/perplexity.py): teacher-forced logits
        ``[B, n_blocks-1, L, V]`` where the context embedding of block i decodes block i+1.

        input_ids / attention_mask ``[B, nb, L]``, block_attention_mask ``[B, nb]`` (see the reference docstring for the
        padding conventions).  Losses (``labels``) are training-only and not implemented.  With
        ``skip_padding_blocks`` the fully padded blocks are not computed and their logits are filled with
        ``eos_token_id`` exactly like the reference (:200-205)."""
        if labels is not None:
            raise NotImplementedError("losses are training-only (out of the decode path); call forward(labels=None)")
        input_ids = input_ids.view((-1,) + input_ids.shape[-2:])
        attention_mask = attention_mask.view((-1,) + attention_mask.shape[-2:])
        block_attention_mask = block_attention_mask.view(-1, block_attention_mask.shape[-1])
        batch_size, n_blocks, block_length = input_ids.shape
        if block_attention_mask.shape != (batch_size, n_blocks):
            raise ValueError("block_attention_mask must have shape (batch_size, n_blocks)"
                             f"expected: {(batch_size, n_blocks)}, got: {block_attention_mask.shape}")
        if n_blocks < 2:
            raise ValueError("forward() needs at least two blocks (block i's context decodes block i+1)")
        td_cfg = self.token_decoder.config
        # EMBEDDING + BLOCK DECODING (all positions)
        input_embeds = self.embedder(input_ids=input_ids.reshape(-1, block_length),
                                     attention_mask=attention_mask.reshape(-1, block_length))
        input_embeds = input_embeds.view(batch_size, n_blocks * self.n_embedding_tokens, input_embeds.shape[-1])
        hidden = self.block_decoder(inputs_embeds=input_embeds, block_attention_mask=block_attention_mask,
                                    return_dict=True).hidden_states
        # TOKEN DECODING: shift blocks left, drop the last block's context
        tok = input_ids[:, 1:, :].reshape(-1, block_length)
        tmask = attention_mask[:, 1:, :].reshape(-1, block_length)
        bmask = block_attention_mask[:, 1:].reshape(-1)
        block_embeddings = hidden[:, :-self.n_embedding_tokens, :].reshape(-1, self.n_embedding_tokens,
                                                                          self.projection_hidden_size)
        keep = None
        if skip_padding_blocks:
            keep = bmask.bool()
            tok, tmask, block_embeddings = tok[keep], tmask[keep], block_embeddings[keep].contiguous()
        local_bs = tok.shape[0]
        vocab = td_cfg.vocab_size
        logits_shape = (batch_size * (n_blocks - 1), block_length, vocab)
        if local_bs > 0:
            prefix = tok.new_full((local_bs, 1), td_cfg.bos_token_id)
            td_ids = torch.cat([prefix, tok], dim=1)
            td_mask = torch.cat([tmask.new_full((local_bs, 1), 1), tmask], dim=1)
            td_logits = self.token_decoder(input_ids=td_ids, attention_mask=td_mask, block_embeddings=block_embeddings,
                                           discard_redundant_tokens=True).logits
        if keep is not None:
            logits = hidden.new_full(logits_shape, fill_value=td_cfg.eos_token_id)
            if local_bs > 0:
                logits[keep] = td_logits
        else:
            logits = td_logits.view(logits_shape)
        logits = logits.view(batch_size, n_blocks - 1, block_length, vocab)
        self.loss = self.token_decoding_loss = self.block_decoding_loss = self.auto_encoding_loss = None
        return {"logits": logits, "loss": None, "block_decoding_loss": None, "token_decoding_loss": None,
                "auto_encoding_loss": None}


class _BlockStreamer:
    """Adapts the engine's per-token callback to the HF streamer protocol the reference relies on: per block
    ``put(placeholder [B,1])``, then ``put(tokens [B])`` per token, then ``end()`` (inference_demo.py:69-83)."""

    def __init__(self, streamer, batch: int, eos_id: int):
        self.s, self.B, self.eos, self.n = streamer, batch, eos_id, 0

    def put(self, t: torch.Tensor):
        self.s.put(t.cpu())

    def block_start(self, n=None, rows=None):
        """``n``: rows the token decoder runs on in this block = the reference's ``decode_mask`` rows (:389-398): the sequences that
        were unfinished when the block started, in ascending original row order -- before and after a compaction alike.  ``rows``:
        their ORIGINAL batch row indices; published as ``streamer.bt_rows`` (LongTensor on the CPU) before the placeholder is put,
        so a batch streamer can attribute the rows of every later ``put`` of the block (the HF protocol itself has no row
        identity; the reference's streamer sees the same compacted rows without one)."""
        if rows is not None:
            try:
                self.s.bt_rows = rows.detach().cpu()
            except AttributeError:      # a streamer with __slots__: the protocol works without the attribution
                pass
        self.s.put(torch.full((self.B if n is None else n, 1), self.eos, dtype=torch.long))

    def block_end(self):
        self.s.end()
