"""CPU ORACLE — test infrastructure, NOT product code.

A pure-PyTorch (CPU, eager) restatement of the reference's Block Transformer decode path. Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product package
(``block_transformer_amd``) never does, and fails loudly when its HIP extension is missing.

Parity status: PINNED against outputs of the reference's own classes imported in the build container
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``: G1 embedder, G2 block-decoder prefill, G4 expansion, G5
token-decoder teacher-forced logits, G6 greedy tokens assembled from the reference classes (arange positions), G7
``BlockTransformer.forward`` logits, G8 preprocess/revert integer cases, G9 greedy tokens from the reference classes
evaluated at the position ids of cached generation).  The reference's cached ``generate()`` itself cannot run on the
installed transformers (5.15.0 vs the pinned 4.39.3): the position ids it would feed
(``hf439_generate`` below: prefix rows at 0, token t at t+1) are restated from the published 4.39.3
``GPTNeoXForCausalLM.prepare_inputs_for_generation`` (position_ids = cumsum(attention_mask) - 1, last column once a cache
exists) and the reference's **kwargs routing (model/token_decoder/gpt_neo_x.py:76-90, base.py:158,301-303); what IS
pinned is the arithmetic of the reference's modules AT those position ids (G9) and that cached == no-cache (G3 property).

Each function cites the reference file:line it follows (paths relative to /root/reference).
The arithmetic lives in third-party ``transformers==4.39.3`` (requirements.txt:3) GPT-NeoX; it is restated from the
published algorithm in ``neox_layer`` below and checked against the installed HF 5.15 GPTNeoXModel through the
reference classes by ``oracle/make_golden.py``.
"""
from __future__ import annotations

import math
import warnings
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
KV = List[Tuple[Tensor, Tensor]]


def td_position_table(policy: str, n_exp: int, block_length: int) -> List[int]:
    """RoPE positions of the adapted token-decoder sequence [BE_1..BE_E, t_0, .., t_{L-2}].

    ``train``: HF default ``arange`` over the adapted sequence (a direct forward without position_ids; reference
    model/token_decoder/base.py:461-502 builds inputs_embeds=[BE.., A, B, C]).
    ``hf439_generate`` (what the reference's generate() computes, the default of ``OracleBlockTransformer.generate``):
    transformers 4.39.3 ``GPTNeoXForCausalLM.prepare_inputs_for_generation`` derives position_ids from the *un-adapted*
    attention mask (cumsum-1) and the reference forwards it through **kwargs (model/token_decoder/base.py:158,301-303 ->
    gpt_neo_x.py:59-71): the E prefix rows all get position 0 (a [B,1] id broadcast over q = E rows), token t gets t+1.
    """
    n = n_exp + block_length - 1
    if policy == "train":
        return list(range(n))
    if policy == "hf439_generate":
        return [0] * n_exp + list(range(1, block_length))
    raise ValueError(f"unknown RoPE position policy {policy!r}")


def rope_tables(max_pos: int, rot: int, base: float, dtype=torch.float32) -> Tuple[Tensor, Tensor]:
    """HF GPTNeoXRotaryEmbedding: inv_freq_j = base^(-2j/rot); angle = pos * inv_freq; returns cos,sin [max_pos, rot/2]."""
    inv_freq = 1.0 / (base ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    t = torch.arange(max_pos, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    return freqs.cos().to(dtype), freqs.sin().to(dtype)


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor, rot: int) -> Tensor:
    """x [B, heads, q, d]; cos/sin [q, rot/2] (already gathered at the positions), or [B, q, rot/2].

    HF apply_rotary_pos_emb with rotate_half on the first ``rot`` dims: pairs (j, j+rot/2).
    """
    half = rot // 2
    x1, x2, rest = x[..., :half], x[..., half:rot], x[..., rot:]
    if cos.dim() == 2:
        c, s = cos[None, None], sin[None, None]
    else:
        c, s = cos[:, None], sin[:, None]
    o1 = x1 * c - x2 * s
    o2 = x2 * c + x1 * s
    return torch.cat([o1, o2, rest], dim=-1)


class OracleDecoder:
    """One GPT-NeoX stack addressed by the reference's state-dict prefix (``block_decoder.`` / ``token_decoder.``)."""

    def __init__(self, sd: Dict[str, Tensor], prefix: str, heads: int, rotary_pct: float = 0.25,
                 rope_base: float = 10000.0, eps: float = 1e-5):
        self.sd, self.p = sd, prefix + "gpt_neox."
        self.heads, self.eps, self.rope_base = heads, eps, rope_base
        self.hidden = sd[self.p + "final_layer_norm.weight"].shape[0]
        self.d = self.hidden // heads
        self.rot = int(self.d * rotary_pct)
        n = 0
        while f"{self.p}layers.{n}.input_layernorm.weight" in sd:
            n += 1
        self.layers = n
        self._tables: Optional[Tuple[Tensor, Tensor]] = None

    def tables(self, max_pos: int, dtype) -> Tuple[Tensor, Tensor]:
        if self._tables is None or self._tables[0].shape[0] < max_pos or self._tables[0].dtype != dtype:
            self._tables = rope_tables(max(max_pos, 64), self.rot, self.rope_base, dtype)
        return self._tables

    def w(self, i: int, name: str) -> Tensor:
        return self.sd[f"{self.p}layers.{i}.{name}"]

    def neox_layer(self, i: int, x: Tensor, positions: Tensor, past: Optional[Tuple[Tensor, Tensor]],
                   key_mask: Optional[Tensor]) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
        """HF GPTNeoXLayer/GPTNeoXAttention/GPTNeoXMLP math:

        a = LN1(x); qkv = a Wqkv^T + b viewed [B,q,heads,3d] (per-head interleaved q|k|v); partial RoPE;
        KV growth by torch.cat (C5); s = softmax(q k^T d^-1/2 + causal + key mask); o = (s v) Wd^T + b;
        m = W2 gelu_erf(W1 LN2(x) + b1) + b2;  x <- m + o + x  (use_parallel_residual).
        """
        B, q, H = x.shape
        heads, d = self.heads, self.d
        a = F.layer_norm(x, (H,), self.w(i, "input_layernorm.weight"), self.w(i, "input_layernorm.bias"), self.eps)
        qkv = F.linear(a, self.w(i, "attention.query_key_value.weight"), self.w(i, "attention.query_key_value.bias"))
        qkv = qkv.view(B, q, heads, 3 * d)
        qh = qkv[..., :d].permute(0, 2, 1, 3)
        kh = qkv[..., d:2 * d].permute(0, 2, 1, 3)
        vh = qkv[..., 2 * d:].permute(0, 2, 1, 3)
        cos_t, sin_t = self.tables(int(positions.max().item()) + 1, x.dtype)
        cos, sin = cos_t[positions], sin_t[positions]
        qh = apply_rope(qh, cos, sin, self.rot)
        kh = apply_rope(kh, cos, sin, self.rot)
        if past is not None:
            kh = torch.cat((past[0], kh), dim=-2)
            vh = torch.cat((past[1], vh), dim=-2)
        n = kh.shape[-2]
        scores = torch.matmul(qh, kh.transpose(-1, -2)) * (d ** -0.5)
        causal = torch.ones(n, n, dtype=torch.bool).tril()[n - q:n, :n]
        neg = torch.finfo(scores.dtype).min
        scores = torch.where(causal[None, None], scores, torch.full_like(scores, neg))
        if key_mask is not None:
            scores = scores.masked_fill(~key_mask.bool()[:, None, None, :n], neg)
        probs = torch.softmax(scores, dim=-1)
        o = torch.matmul(probs, vh).permute(0, 2, 1, 3).reshape(B, q, H)
        o = F.linear(o, self.w(i, "attention.dense.weight"), self.w(i, "attention.dense.bias"))
        m = F.layer_norm(x, (H,), self.w(i, "post_attention_layernorm.weight"),
                         self.w(i, "post_attention_layernorm.bias"), self.eps)
        m = F.linear(m, self.w(i, "mlp.dense_h_to_4h.weight"), self.w(i, "mlp.dense_h_to_4h.bias"))
        m = F.gelu(m)
        m = F.linear(m, self.w(i, "mlp.dense_4h_to_h.weight"), self.w(i, "mlp.dense_4h_to_h.bias"))
        return m + o + x, (kh, vh)

    def forward(self, x: Tensor, positions: Tensor, past: Optional[KV] = None,
                key_mask: Optional[Tensor] = None) -> Tuple[Tensor, KV]:
        """HF GPTNeoXModel.forward with inputs_embeds: NL layers then final_layer_norm."""
        new_past: KV = []
        for i in range(self.layers):
            x, kv = self.neox_layer(i, x, positions, None if past is None else past[i], key_mask)
            new_past.append(kv)
        H = x.shape[-1]
        x = F.layer_norm(x, (H,), self.sd[self.p + "final_layer_norm.weight"],
                         self.sd[self.p + "final_layer_norm.bias"], self.eps)
        return x, new_past


class OracleBlockTransformer:
    """Restatement of BlockTransformer + LookupEmbedder + GPTNeoXBlockDecoder + GPTNeoXTokenDecoder (prefix strategy)."""

    def __init__(self, state_dict: Dict[str, Tensor], block_length: int, bd_heads: int, td_heads: int,
                 expansion_ratio: int = 2, rotary_pct: float = 0.25, rope_base: float = 10000.0, eps: float = 1e-5,
                 eos_token_id: int = 0, pad_token_id: Optional[int] = None, gen_eos_token_id: Optional[int] = None,
                 gen_pad_token_id: Optional[int] = None, embedder_pad_token_id: Optional[int] = None,
                 rope_policy: str = "hf439_generate"):
        self.sd = state_dict
        self.L, self.E = block_length, expansion_ratio
        self.bd = OracleDecoder(state_dict, "block_decoder.", bd_heads, rotary_pct, rope_base, eps)
        self.td = OracleDecoder(state_dict, "token_decoder.", td_heads, rotary_pct, rope_base, eps)
        self.H = self.bd.hidden
        self.Ht = self.td.hidden
        self.vocab = state_dict["token_decoder.embed_out.weight"].shape[0]
        # token ids: reference model/token_decoder/base.py:53-54,73-74 (bos/pad default to eos)
        self.eos_token_id = eos_token_id
        self.bos_token_id = eos_token_id
        self.pad_token_id = eos_token_id if pad_token_id is None else pad_token_id
        self.gen_eos_token_id = gen_eos_token_id          # token_decoder.generation_config.eos_token_id
        # HF generate: pad_token_id None + eos set -> pad = eos
        self.gen_pad_token_id = gen_pad_token_id if gen_pad_token_id is not None else (
            gen_eos_token_id if gen_eos_token_id is not None else self.pad_token_id)
        self.embedder_pad_token_id = embedder_pad_token_id  # LookupConfig.pad_token_id (None, Q4)
        self.rope_policy = rope_policy
        self.td_pos = td_position_table(rope_policy, self.E, self.L)   # cached generation (token_decoder_generate)

    # ---- embedder -------------------------------------------------------------------------------------------
    def embed(self, ids: Tensor) -> Tensor:
        """LookupEmbedder.forward (model/embedder/lookup.py:71-87) + concat projection (embedder/base.py:79-89).

        ids [..., L] -> [..., H]: out[l*h_e + j] = W_emb[ids[l], j]; the attention mask is ignored (Q4)."""
        w = self.sd["embedder.embeddings.weight"]
        ids = ids.reshape(-1, ids.shape[-1])
        if ids.shape[-1] != self.L:
            raise ValueError(f"Sequence length must be {self.L}, but got {ids.shape[-1]}")
        return w[ids].reshape(ids.shape[0], -1)

    # ---- block decoder --------------------------------------------------------------------------------------
    def block_decoder(self, x: Tensor, block_mask: Tensor, past: Optional[KV] = None) -> Tuple[Tensor, KV]:
        """GPTNeoXBlockDecoder.forward (model/block_decoder/gpt_neo_x.py:71-126): positions arange(past, past+q)."""
        past_len = 0 if past is None else past[0][0].shape[-2]
        q = x.shape[1]
        positions = torch.arange(past_len, past_len + q)
        return self.bd.forward(x, positions, past, block_mask)

    # ---- token decoder --------------------------------------------------------------------------------------
    def expand(self, ctx: Tensor) -> Tensor:
        """expand_block_embeddings (model/token_decoder/base.py:118-139): Conv1d(k=1) == Linear; [B,H]->[B,E,Ht]."""
        w = self.sd["token_decoder.expansion_layer.weight"][:, :, 0]
        b = self.sd["token_decoder.expansion_layer.bias"]
        return F.linear(ctx, w, b).reshape(ctx.shape[0], self.E, self.Ht)

    def logits(self, h: Tensor) -> Tensor:
        return F.linear(h, self.sd["token_decoder.embed_out.weight"])

    def token_decoder_teacher_forced(self, ctx: Tensor, tokens: Tensor, token_mask: Optional[Tensor] = None,
                                     positions=None) -> Tensor:
        """BaseTokenDecoder.forward(past=None) with input_ids=[BOS, tokens...] (base.py:141-397, prefix adaptation
        :461-502, redundant-prefix drop :321-329): returns logits [B, 1+len(tokens), V] (one per input position)."""
        B = ctx.shape[0]
        emb_in = self.sd["token_decoder.gpt_neox.embed_in.weight"]
        x = torch.cat([self.expand(ctx), emb_in[tokens]], dim=1)
        n = x.shape[1]
        # a direct forward carries no position_ids: HF default arange (the reference's training / scoring path)
        positions = torch.arange(n) if positions is None else torch.as_tensor(positions)[:n]
        key_mask = None
        if token_mask is not None:
            key_mask = torch.cat([torch.ones(B, self.E, dtype=token_mask.dtype), token_mask], dim=1)
        h, _ = self.td.forward(x, positions, None, key_mask)
        return self.logits(h[:, self.E - 1:])

    def token_decoder_generate(self, ctx: Tensor, n_tokens: int, return_logits: bool = False):
        """BaseTokenDecoder.generate (base.py:399-453) -> HF greedy_search with the local KV cache.

        step 0: inputs_embeds=[BE_1..BE_E] (q=E), keep hidden[:, E-1:]; step t>=1: embed_in(token) (q=1).
        HF greedy_search semantics: once generation_config.eos_token_id is produced by a row, its later tokens are
        generation_config.pad_token_id, and the loop stops right after the step at which every row has produced
        EOS (Q6) - so fewer than n_tokens columns may come back.  Returns tokens [B, <=n_tokens] (without the EOS
        placeholder column)."""
        B = ctx.shape[0]
        emb_in = self.sd["token_decoder.gpt_neox.embed_in.weight"]
        x = self.expand(ctx)
        past: Optional[KV] = None
        toks, all_logits = [], []
        unfinished = torch.ones(B, dtype=torch.long)
        pos0 = 0
        for t in range(n_tokens):
            q = x.shape[1]
            positions = torch.tensor(self.td_pos[pos0:pos0 + q])
            h, past = self.td.forward(x, positions, past, None)
            pos0 += q
            lg = self.logits(h[:, -1])
            all_logits.append(lg)
            nxt = lg.argmax(-1)
            if self.gen_eos_token_id is not None:
                nxt = nxt * unfinished + self.gen_pad_token_id * (1 - unfinished)
                unfinished = unfinished * (nxt != self.gen_eos_token_id).long()
            toks.append(nxt)
            if self.gen_eos_token_id is not None and int(unfinished.max()) == 0:
                break
            x = emb_in[nxt][:, None, :]
        out = torch.stack(toks, dim=1)
        if return_logits:
            return out, torch.stack(all_logits, dim=1)
        return out

    # ---- orchestration --------------------------------------------------------------------------------------
    def preprocess_inputs_for_generation(self, input_ids: Tensor, attention_mask: Optional[Tensor] = None):
        """model/block_transformer.py:227-266 (left pad to a multiple of L; pad id None -> value 0, Q4)."""
        input_ids = input_ids.view(-1, input_ids.shape[-1])
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
            if self.embedder_pad_token_id is not None:
                attention_mask[input_ids == self.embedder_pad_token_id] = 0
        else:
            attention_mask = attention_mask.view(-1, attention_mask.shape[-1])
        bs = input_ids.shape[0]
        rem = input_ids.shape[-1] % self.L
        pad_length = 0
        if rem != 0:
            pad_length = self.L - rem
            pad_val = 0 if self.embedder_pad_token_id is None else self.embedder_pad_token_id
            input_ids = F.pad(input_ids, (pad_length, 0), value=pad_val)
            attention_mask = F.pad(attention_mask, (pad_length, 0), value=0)
        input_ids = input_ids.view(bs, -1, self.L)
        attention_mask = attention_mask.view(bs, -1, self.L)
        block_attention_mask = attention_mask.any(dim=-1).long()
        return dict(input_ids=input_ids, attention_mask=attention_mask, block_attention_mask=block_attention_mask,
                    initial_block_padding=pad_length)

    @staticmethod
    def revert_block_input_ids_to_vanilla(input_ids: Tensor, added_initial_block_padding: int,
                                          last_block_unfilled_length: int) -> Tensor:
        """model/block_transformer.py:268-277."""
        input_ids = input_ids.reshape(-1, input_ids.shape[-2] * input_ids.shape[-1])
        input_ids = input_ids[:, added_initial_block_padding:]
        if last_block_unfilled_length > 0:
            input_ids = input_ids[:, :-last_block_unfilled_length]
        return input_ids

    def generate(self, input_ids: Tensor, attention_mask: Optional[Tensor] = None,
                 block_attention_mask: Optional[Tensor] = None, max_length: int = 100,
                 return_trace: bool = False):
        """BlockTransformer.generate (model/block_transformer.py:279-449), greedy.

        Follows the reference loop including: Q1 (early return = warn + return input unchanged), Q2 (end test
        ignores initial padding), Q5 (finished rows: zero block embedding, mask column 0, pad tokens; 'unfinished'
        needs no EOS anywhere in the block), Q6.  Finished rows are computed and overwritten instead of being
        compacted away (same results)."""
        added_pad = 0
        vanilla_mode = block_attention_mask is None
        if vanilla_mode:
            d = self.preprocess_inputs_for_generation(input_ids, attention_mask)
            input_ids, attention_mask = d["input_ids"], d["attention_mask"]
            block_attention_mask, added_pad = d["block_attention_mask"], d["initial_block_padding"]
        else:
            if input_ids.shape[-1] != self.L:
                raise ValueError("input_ids.shape[-1] must be equal to self.block_length when block_attention_mask "
                                 "is given")
            if attention_mask is None:
                raise ValueError("attention_mask must be specified when block_attention_mask is specified")
        nb, L = input_ids.shape[-2:]
        input_ids = input_ids.reshape(-1, nb, L)
        block_attention_mask = block_attention_mask.reshape(-1, nb)
        B = input_ids.shape[0]
        if nb * L - added_pad > max_length:
            warnings.warn("max_length is less than the length of the input_ids. Returning original input_ids.")
            if vanilla_mode:
                return self.revert_block_input_ids_to_vanilla(input_ids, added_pad, 0)
            return input_ids

        x = self.embed(input_ids).view(B, nb, self.H)
        unfinished = torch.ones(B, dtype=torch.long)
        past: Optional[KV] = None
        trace = dict(ctx=[], tokens=[])
        while True:
            hidden, past = self.block_decoder(x, block_attention_mask, past)
            ctx = hidden[:, -1, :]
            current_length = input_ids.shape[1] * L - added_pad
            next_token_count = min(max_length - current_length, L)
            next_tokens = torch.full((B, L), self.pad_token_id, dtype=torch.long)
            decode_mask = unfinished.bool()
            gen = self.token_decoder_generate(ctx[decode_mask], next_token_count)  # compaction, reference :389-395
            next_tokens[decode_mask, :gen.shape[-1]] = gen
            if return_trace:
                trace["ctx"].append(ctx.clone())
                trace["tokens"].append(next_tokens.clone())
            input_ids = torch.cat([input_ids, next_tokens[:, None, :]], dim=1)
            if self.gen_eos_token_id is not None:
                unfinished = unfinished * (next_tokens != self.gen_eos_token_id).all(dim=-1).long()
            if input_ids.shape[1] * L >= max_length or int(unfinished.sum()) == 0:
                break
            x = torch.zeros(B, 1, self.H, dtype=x.dtype)
            x[decode_mask, 0] = self.embed(next_tokens[decode_mask])
            block_attention_mask = torch.cat([block_attention_mask, unfinished[:, None]], dim=-1)
        out = input_ids
        if vanilla_mode:
            out = self.revert_block_input_ids_to_vanilla(input_ids, added_pad, L - next_token_count)
        if return_trace:
            return out, trace
        return out

    def forward_logits(self, input_ids: Tensor, attention_mask: Tensor, block_attention_mask: Tensor) -> Tensor:
        """BlockTransformer.forward with labels=None, skip_padding_blocks=False-equivalent dense output
        (model/block_transformer.py:50-225): context embedding of block i decodes block i+1.
        Returns logits [B, nb-1, L, V]."""
        B, nb, L = input_ids.shape
        x = self.embed(input_ids).view(B, nb, self.H)
        hidden, _ = self.block_decoder(x, block_attention_mask, None)
        ctx = hidden[:, :-1].reshape(B * (nb - 1), self.H)
        toks = input_ids[:, 1:].reshape(B * (nb - 1), L)
        tmask = attention_mask[:, 1:].reshape(B * (nb - 1), L)
        # discard_redundant_tokens=True drops the last input column: inputs [BOS, A, B, C] -> logits for A^,B^,C^,D^
        lg = self.token_decoder_teacher_forced(ctx, toks[:, :-1], tmask[:, :-1])
        return lg.view(B, nb - 1, L, self.vocab)


def random_state_dict(dims, seed: int = 0, dtype=torch.float32, include_unused_embed_in: bool = True,
                      bias_std: float = 0.0) -> Dict[str, Tensor]:
    """HF-style random init for a ``block_transformer_amd.config.ModelDims``: N(0, 0.02) Linear/Embedding weights, zero
    biases (or N(0, bias_std) to make parity tests sensitive to bias handling), LayerNorm (1, 0), expansion conv
    N(0, H^-1/2) (reference model/token_decoder/base.py:455-459).  Key names are the reference's state_dict names
    (checkpoint row)."""
    g = torch.Generator().manual_seed(seed)

    def nrm(*shape, std=0.02):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    def bias(n):
        return nrm(n, std=bias_std) if bias_std > 0 else torch.zeros(n, dtype=dtype)

    sd: Dict[str, Tensor] = {}
    sd["embedder.embeddings.weight"] = nrm(dims.vocab, dims.emb_hidden)
    for prefix, dd in (("token_decoder.", dims.td), ("block_decoder.", dims.bd)):
        p = prefix + "gpt_neox."
        if prefix == "token_decoder." or include_unused_embed_in:
            sd[p + "embed_in.weight"] = nrm(dims.vocab, dd.hidden)
        for i in range(dd.layers):
            q = f"{p}layers.{i}."
            H, I = dd.hidden, dd.inter
            ln_w = (lambda: torch.ones(H, dtype=dtype)) if bias_std == 0 else (lambda: (1 + nrm(H, std=bias_std)))
            sd[q + "input_layernorm.weight"] = ln_w()
            sd[q + "input_layernorm.bias"] = bias(H)
            sd[q + "post_attention_layernorm.weight"] = ln_w()
            sd[q + "post_attention_layernorm.bias"] = bias(H)
            sd[q + "attention.query_key_value.weight"] = nrm(3 * H, H)
            sd[q + "attention.query_key_value.bias"] = bias(3 * H)
            sd[q + "attention.dense.weight"] = nrm(H, H)
            sd[q + "attention.dense.bias"] = bias(H)
            sd[q + "mlp.dense_h_to_4h.weight"] = nrm(I, H)
            sd[q + "mlp.dense_h_to_4h.bias"] = bias(I)
            sd[q + "mlp.dense_4h_to_h.weight"] = nrm(H, I)
            sd[q + "mlp.dense_4h_to_h.bias"] = bias(H)
        sd[p + "final_layer_norm.weight"] = torch.ones(dd.hidden, dtype=dtype) if bias_std == 0 else 1 + nrm(dd.hidden, std=bias_std)
        sd[p + "final_layer_norm.bias"] = bias(dd.hidden)
    sd["token_decoder.embed_out.weight"] = nrm(dims.vocab, dims.td.hidden)
    sd["token_decoder.expansion_layer.weight"] = nrm(dims.E * dims.td.hidden, dims.bd.hidden, 1,
                                                     std=dims.bd.hidden ** -0.5)
    sd["token_decoder.expansion_layer.bias"] = bias(dims.E * dims.td.hidden)
    return sd


def oracle_from_dims(dims, sd: Dict[str, Tensor], **kw) -> OracleBlockTransformer:
    return OracleBlockTransformer(sd, dims.block_length, dims.bd.heads, dims.td.heads, dims.expansion_ratio,
                                  dims.bd.rotary_pct, dims.bd.rope_base, dims.bd.ln_eps, **kw)
