"""GPTNeoXTokenDecoder — same constructor / attribute paths / forward + generate contract as the reference's
model/token_decoder/gpt_neo_x.py:34-90 and model/token_decoder/base.py:118-502 (prefix strategy), with the arithmetic in
libbt_hip.so.

forward(input_ids=[[BOS, A, B, ...]], block_embeddings=[B, 1, H_bd])          teacher-forced / first step
forward(input_ids=[[t]], expanded_block_embeddings=..., past_key_values=c)   cached step
    -> .logits [B, n_out, V], .past_key_values (local KVCache)
generate(block_embeddings=[B, 1, H_bd], max_length=k+1) -> [B, <= k+1] ids, column 0 = EOS placeholder."""
from __future__ import annotations

import warnings
from typing import Optional

import torch
from torch import nn

from ... import _lib, ops
from ...engine import DEFAULT_ROPE_POLICY, TokenBlockState, td_position_table
from ...runtime import DecoderRuntime, KVCache
from ..neox import CausalLMOutputWithPast, GenerationConfig, GPTNeoXModel, init_neox_weights
from .base import BaseTokenDecoder


class GPTNeoXTokenDecoder(nn.Module, BaseTokenDecoder):
    def __init__(self, config, block_length, n_embedding_tokens=None, projection_hidden_size=None,
                 expansion_method=None, expansion_ratio: int = 1, init_expansion=True, decoding_strategy=None):
        nn.Module.__init__(self)
        self.config = config
        self.gpt_neox = GPTNeoXModel(config)
        self.embed_out = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        init_neox_weights(self, config.initializer_range)
        pad_at_init = config.pad_token_id  # HF builds generation_config before BaseTokenDecoder defaults pad := eos
        BaseTokenDecoder.__init__(self, config, block_length, n_embedding_tokens, projection_hidden_size,
                                  expansion_method, expansion_ratio, init_expansion, decoding_strategy)
        # HF GenerationConfig.from_model_config: eos/bos/pad copied from the model config at construction
        self.generation_config = GenerationConfig(eos_token_id=config.eos_token_id, bos_token_id=config.bos_token_id,
                                                  pad_token_id=pad_at_init)
        # RoPE positions used by generate() (engine.td_position_table): the reference's cached generation rotates the E
        # prefix rows at position 0 and token t at t + 1 (transformers 4.39.3 position_ids routed through **kwargs);
        # "train" = plain arange.  A direct forward() without ``position_ids`` always uses arange, like the reference.
        self.rope_policy = DEFAULT_ROPE_POLICY
        self._rt = DecoderRuntime(self.gpt_neox, config)
        self._tok_state: Optional[TokenBlockState] = None
        self._pos_tables = {}

    # ---- reference adapter surface (model/token_decoder/gpt_neo_x.py:52-74) ------------------------------------------
    @torch.no_grad()
    def embed_input_ids(self, input_ids: torch.LongTensor) -> torch.Tensor:
        return ops.embed_rows(input_ids, self.gpt_neox.embed_in.weight)

    @torch.no_grad()
    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return ops.linear(hidden_states, self.embed_out.weight)

    @torch.no_grad()
    def expand_block_embeddings(self, block_embeddings: torch.Tensor) -> torch.Tensor:
        """[B, 1, H_bd] -> [B, E, H]   (model/token_decoder/base.py:118-139; Conv1d(k=1) == Linear)."""
        B = block_embeddings.shape[0]
        y = ops.linear(block_embeddings.reshape(B, -1), self.expansion_layer.weight, self.expansion_layer.bias)
        return y.view(B, self.n_expanded_emb, self.hidden_size)

    def _pos_table_cached(self, table, device) -> torch.Tensor:
        """int32 device copy of a RoPE position table, built once per distinct table (no per-call H2D copy / sync)."""
        key = (table, str(device))
        t = self._pos_tables.get(key)
        if t is None:
            if len(self._pos_tables) > 64:
                self._pos_tables.clear()
            t = self._pos_tables[key] = torch.tensor(list(table), dtype=torch.int32, device=device)
        return t

    # ---- forward ----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_ids: torch.LongTensor, attention_mask: Optional[torch.Tensor] = None,
                block_embeddings: Optional[torch.Tensor] = None, past_key_values: Optional[KVCache] = None,
                labels=None, return_dict: Optional[bool] = None,
                expanded_block_embeddings: Optional[torch.Tensor] = None, discard_redundant_tokens: bool = False,
                log_loss_by_position=True, **kwargs):
        if labels is not None:
            raise NotImplementedError("token-decoder loss is training-only (out of the decode path)")
        cfg = self.config
        E, H = self.n_expanded_emb, self.hidden_size
        input_ids = input_ids.view(-1, input_ids.shape[-1])
        bs = input_ids.shape[0]
        if past_key_values is None and bool((input_ids[:, 0] != cfg.bos_token_id).any()):
            raise ValueError(f"input_ids must start with BOS token {cfg.bos_token_id}, got {input_ids[:, 0]}. "
                             f"Check the docstrings")
        if self.block_length is not None and input_ids.shape[1] > self.block_length + 1:
            warnings.warn(f"input_ids.shape[1] should be <= block_length + 1, got {input_ids.shape[1]}. "
                          f"You are attempting to infer beyond `block_length` on which the model was trained.")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        else:
            attention_mask = attention_mask.view(-1, attention_mask.shape[-1])
            if bool((attention_mask[:, 0] != 1).any()):
                raise ValueError("attention_mask[:, 0] must be 1")
        past_len = 0 if past_key_values is None else past_key_values.length
        if sum([block_embeddings is None, expanded_block_embeddings is None]) != 1:
            raise ValueError("block_embeddings xor expanded_block_embeddings must be specified")
        if block_embeddings is None:
            if expanded_block_embeddings.shape != (bs, E, H):
                raise ValueError(f"expanded_block_embeddings must be of shape (batch_size, n_expanded_emb, "
                                 f"hidden_size) = ({bs}, {E}, {H}), got {expanded_block_embeddings.shape}")
        else:
            if past_key_values is not None:
                raise ValueError("you should pass the pre-expanded `expanded_block_embeddings` instead of "
                                 "`block_embeddings` to prevent redundant computation during generation. "
                                 "past_key_values is typically used for generation.")
            expanded_block_embeddings = self.expand_block_embeddings(
                block_embeddings.view(bs, -1, block_embeddings.shape[-1]))
        redundant = None
        if discard_redundant_tokens:
            content = (input_ids != cfg.pad_token_id) | (attention_mask == 1)
            input_ids = input_ids[:, :-1]
            attention_mask = attention_mask[:, :-1]
            redundant = ~content[:, 1:]
        # prefix adaptation (base.py:461-502)
        if past_len == 0:
            tokens = input_ids[:, 1:]
            x = torch.empty(bs, E + tokens.shape[1], H, dtype=self._rt.dtype, device=expanded_block_embeddings.device)
            x[:, :E] = expanded_block_embeddings
            if tokens.shape[1] > 0:
                x[:, E:] = self.embed_input_ids(tokens.contiguous())
            key_mask = torch.cat([torch.ones(bs, E, dtype=torch.uint8, device=x.device),
                                  attention_mask[:, 1:].to(torch.uint8)], dim=1)
            first_out = E - 1
        else:
            x = self.embed_input_ids(input_ids.contiguous())
            key_mask = torch.cat([torch.ones(bs, E, dtype=torch.uint8, device=x.device),
                                  attention_mask[:, 1:].to(torch.uint8)], dim=1)
            first_out = 0
        q = x.shape[1]
        n_total = past_len + q
        if key_mask.shape[1] != n_total:
            raise ValueError(f"attention_mask covers {key_mask.shape[1]} adapted positions, expected {n_total}")
        cache = past_key_values
        heads = cfg.num_attention_heads
        if cache is None:
            cache = KVCache(cfg.num_hidden_layers, bs, heads, max(n_total, E + self.block_length - 1), H // heads,
                            x.dtype, x.device)
        elif n_total > cache.n_max:
            cache = cache.grown(n_total + self.block_length)
        # RoPE positions of the adapted rows: HF default arange(past, past + q) unless the caller passes ``position_ids``
        # (the reference forwards **kwargs to gpt_neox, base.py:301-303; HF broadcasts a [B, 1] id over the q rows)
        position_ids = kwargs.get("position_ids")
        table = list(range(cache.n_max))
        if position_ids is not None:
            pid = torch.as_tensor(position_ids).reshape(-1, torch.as_tensor(position_ids).shape[-1]).cpu()
            if not bool((pid == pid[:1]).all()):
                raise NotImplementedError("per-sequence position_ids are not supported: all rows must share one position row")
            row = [int(v) for v in pid[0]]
            if len(row) == 1:
                row = row * q
            if len(row) != q:
                raise ValueError(f"position_ids has {len(row)} columns, expected 1 or the {q} adapted rows")
            table[past_len:past_len + q] = row
        pos_table = self._pos_table_cached(tuple(table), x.device)
        hidden = self._rt.forward(x.view(bs * q, H), q, cache, pos0=past_len, pos_table=pos_table,
                                  mask=key_mask.contiguous(), first_out_row=first_out,
                                  rope=self._rt.rope(max(table) + 1))
        self._rt.check_errors()
        cache.length = n_total
        n_out = q - first_out
        logits = self.compute_logits(hidden).view(bs, n_out, cfg.vocab_size)
        if redundant is not None:
            logits = logits.masked_fill(redundant[:, :n_out, None], 0)
        return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=cache, hidden_states=None,
                                      attentions=None)

    # ---- generate -----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids: Optional[torch.LongTensor] = None, block_embeddings: torch.Tensor = None,
                 attention_mask: Optional[torch.LongTensor] = None, max_length: Optional[int] = None,
                 max_new_tokens: Optional[int] = None, streamer=None, **kwargs):
        """model/token_decoder/base.py:399-453 + HF greedy_search / sample semantics (EOS -> pad afterwards, stop once
        every row has produced EOS)."""
        if block_embeddings is None:
            raise ValueError("block_embeddings must be specified for generation with token decoder")
        block_embeddings = block_embeddings.view(-1, block_embeddings.shape[-2], block_embeddings.shape[-1])
        bs = block_embeddings.shape[0]
        cfg = self.config
        if input_ids is None:
            input_ids = torch.full((bs, 1), cfg.eos_token_id, dtype=torch.long, device=block_embeddings.device)
            if attention_mask is not None:
                raise ValueError("attention_mask must be None when input_ids is None")
        input_ids = input_ids.view(-1, input_ids.shape[-1])
        if input_ids.shape[1] != 1:
            raise NotImplementedError("token-decoder generation starts from the single EOS placeholder")
        if max_new_tokens:
            max_length = input_ids.shape[1] + max_new_tokens
        if max_length is None:
            max_length = self.block_length + 1
        elif max_length > self.block_length + 1:
            raise NotImplementedError(f"max_length {max_length} > block_length + 1: the local window holds "
                                      f"{self.n_expanded_emb + self.block_length - 1} positions")
        if bool((input_ids[:, 0] != cfg.eos_token_id).any()):
            raise ValueError(f"input_ids must start with EOS token {cfg.eos_token_id}, got {input_ids[:, 0]}. "
                             f"Check the docstrings")
        count = max_length - 1
        st = self._tok_state
        if st is None or st.B != bs or st.dtype != self._rt.dtype or st.local.k.device != block_embeddings.device \
                or st.policy != self.rope_policy:
            st = self._tok_state = TokenBlockState(self, bs, block_embeddings.device, self.rope_policy)
        gen = self.generation_config
        if streamer is not None:
            streamer.put(input_ids.cpu())
        on_token = (lambda t: streamer.put(t.cpu())) if streamer is not None else None
        raw = st.run(block_embeddings.reshape(bs, -1).contiguous(), count, gen, on_token)[:, :count].clone()
        if streamer is not None:
            streamer.end()
        if gen.eos_token_id is not None:
            pad = gen.pad_token_id if gen.pad_token_id is not None else gen.eos_token_id
            unfinished = torch.ones(bs, dtype=torch.long, device=raw.device)
            cols = []
            for t in range(count):
                nxt = raw[:, t] * unfinished + pad * (1 - unfinished)
                unfinished = unfinished * (nxt != gen.eos_token_id).long()
                cols.append(nxt)
                if int(unfinished.max()) == 0:
                    break
            raw = torch.stack(cols, dim=1)
        return torch.cat([input_ids, raw], dim=1)
