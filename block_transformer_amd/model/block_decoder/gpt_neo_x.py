"""GPTNeoXBlockDecoder — same constructor / attribute paths / forward contract as the reference's
model/block_decoder/gpt_neo_x.py:56-126, with the GPT-NeoX stack executed by libbt_hip.so.

forward(inputs_embeds=[B, q, H], block_attention_mask=[B, n_total], past_key_values=cache|None) ->
    .hidden_states [B, q, H] (after final_layer_norm), .past_key_values (KVCache), .loss None, .logits None

* past_key_values is None: PREFILL over q block positions (bf16: MFMA flash kernel; fp32: row-wise kernel).
* past_key_values given:    appends q rows at positions past_len.. (decode step when q == 1).
Positions are arange(past_len, past_len + q) as in HF GPTNeoXModel; only n_embedding_tokens == 1 is supported (the
reference asserts the same for its flash path, :30, and for generation, block_transformer.py:365-366)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from ... import _lib
from ...runtime import DecoderRuntime, KVCache
from ..neox import CausalLMOutputWithPast, GPTNeoXModel, init_neox_weights
from .base import BaseBlockDecoder

PREFILL_ROWS = 32768  # rows per prefill chunk (bounds the activation workspace: 9*H elements per row)


class GPTNeoXBlockDecoder(nn.Module, BaseBlockDecoder):
    def __init__(self, config, n_embedding_tokens=8, use_block_decoding_loss=False,
                 block_decoding_loss_type="contrastive", keep_unused_embed_in: bool = True):
        nn.Module.__init__(self)
        if n_embedding_tokens != 1:
            raise NotImplementedError("generation currently only implemented for n_embedding_tokens == 1")
        if use_block_decoding_loss:
            raise NotImplementedError("block decoding loss is training-only (out of the decode path)")
        self.config = config
        # the reference allocates gpt_neox.embed_in [V, H] but never uses it (gpt_neo_x.py:53, Q8): kept (optionally)
        # only so checkpoints load with strict=True
        self.gpt_neox = GPTNeoXModel(config, with_embed_in=keep_unused_embed_in)
        self.embed_out = None
        BaseBlockDecoder.__init__(self, config.hidden_size, n_embedding_tokens, use_block_decoding_loss,
                                  block_decoding_loss_type)
        init_neox_weights(self, config.initializer_range)
        self._rt = DecoderRuntime(self.gpt_neox, config)

    @torch.no_grad()
    def forward(self, input_ids=None, block_attention_mask: Optional[torch.Tensor] = None, position_ids=None,
                inputs_embeds: Optional[torch.Tensor] = None, head_mask=None,
                past_key_values: Optional[KVCache] = None, labels=None, use_cache: Optional[bool] = None,
                output_attentions=None, output_hidden_states=None, return_dict: Optional[bool] = None):
        if input_ids is not None:
            raise ValueError("block decoders do not use input_ids")
        if labels is not None:
            raise ValueError("block decoders do not use labels")
        if position_ids is not None or head_mask is not None or output_attentions:
            raise NotImplementedError("position_ids / head_mask / output_attentions are not part of the decode path")
        if inputs_embeds is None or inputs_embeds.dim() != 3:
            raise ValueError("inputs_embeds must be [batch, n_blocks, hidden_size]")
        B, q, H = inputs_embeds.shape
        rt = self._rt
        cache = past_key_values
        pos0 = 0 if cache is None else cache.length
        n_total = pos0 + q
        cfg = self.config
        if cache is None:
            heads = cfg.num_attention_heads
            cache = KVCache(cfg.num_hidden_layers, B, heads, ((n_total + 63) // 64) * 64 + 64, H // heads,
                            inputs_embeds.dtype, inputs_embeds.device)
        elif n_total > cache.n_max:
            cache = cache.grown(max(2 * cache.n_max, n_total + 64))
        if cache.batch != B:
            raise ValueError(f"past_key_values were built for batch {cache.batch}, got {B}")
        mask = None
        if block_attention_mask is not None:
            if block_attention_mask.shape != (B, n_total):
                raise ValueError(f"block_attention_mask must have shape {(B, n_total)}, got "
                                 f"{tuple(block_attention_mask.shape)}")
            mask = block_attention_mask.to(torch.uint8).contiguous()
        mode = _lib.ATTN_FLASH if (inputs_embeds.dtype == torch.bfloat16 and q > 1) else _lib.ATTN_ROWS
        out = torch.empty(B, q, H, dtype=inputs_embeds.dtype, device=inputs_embeds.device)
        bc = max(1, min(B, PREFILL_ROWS // q))
        for b0 in range(0, B, bc):
            b1 = min(B, b0 + bc)
            x = inputs_embeds[b0:b1].reshape((b1 - b0) * q, H).clone()  # residual stream is updated in place
            rt.forward(x, q, cache, b0=b0, pos0=pos0, mask=None if mask is None else mask[b0:b1], mode=mode,
                       out=out[b0:b1].view(-1, H))
        rt.check_errors()
        cache.length = n_total
        return CausalLMOutputWithPast(loss=None, logits=None, past_key_values=cache, hidden_states=out,
                                      attentions=None)
