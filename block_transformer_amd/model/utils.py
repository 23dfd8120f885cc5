"""Factories with the reference's names and argument meaning (model/utils.py:84-228), building the MI355X modules.

``cfg`` may be a plain dict (block_transformer_amd.config.load_config) or any attribute-style config object (e.g. an
OmegaConf DictConfig if the caller has one).  Differences from the reference, all forced by the environment:
* no hub access: ``model_name_or_path`` is not fetched; the pythia-410m-deduped base values it would deliver are the
  defaults of ``GPTNeoXConfig`` (config.PYTHIA_BASE);
* ``use_pretrained_weights: true`` is rejected (load a local safetensors checkpoint instead, see inference_demo.py);
* the tokenizer is not downloaded: ``load_block_transformer_from_config`` returns ``(model, None)`` unless a local
  tokenizer directory is given via ``cfg.tokenizer.local_path``."""
from __future__ import annotations

import warnings
from collections.abc import Mapping
from typing import Any, Optional, Tuple

import torch

from ..config import head_dim_rule
from .block_decoder.gpt_neo_x import GPTNeoXBlockDecoder
from .block_transformer import BlockTransformer
from .embedder.lookup import LookupConfig, LookupEmbedder
from .neox import GPTNeoXConfig
from .token_decoder.gpt_neo_x import GPTNeoXTokenDecoder

EMBEDDER_CLS = {"lookup": LookupEmbedder}
EMBEDDER_CONFIG_CLS = {"lookup": LookupConfig}
BLOCK_DECODER_CLS = {"gpt-neo-x": GPTNeoXBlockDecoder}
TOKEN_DECODER_CLS = {"gpt-neo-x": GPTNeoXTokenDecoder}


def _get(cfg: Any, path: str, default=None):
    node = cfg
    for part in path.split("."):
        if node is None:
            return default
        if isinstance(node, Mapping):
            node = node.get(part, None)
        else:
            node = getattr(node, part, None)
    return default if node is None else node


def get_torch_dtype(cfg) -> torch.dtype:
    return {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[_get(cfg, "precision", "bf16")]


def _neox_config(node, torch_dtype) -> GPTNeoXConfig:
    c = dict(_get(node, "config", {}) or {})
    hidden = int(c["hidden_size"])
    if c.get("num_attention_heads") is None:  # reference util/config.py:92-102 (preprocess_config autofill)
        c["num_attention_heads"] = hidden // head_dim_rule(hidden)
    if c.get("intermediate_size") is None:    # util/config.py:103-105
        c["intermediate_size"] = hidden * 4
    known = GPTNeoXConfig().__dict__
    for k in c:
        if k not in known:
            raise ValueError(f"Config key {k} not found in decoder config.")
    return GPTNeoXConfig(torch_dtype=torch_dtype, attn_implementation=_get(node, "attn_implementation"), **c)


def _cls(table, name, what):
    try:
        return table[name]
    except KeyError:
        raise NotImplementedError(f"{what} cls={name!r} is outside the MI355X decode path (SURVEY.md section 2, row 17): "
                                  f"supported: {sorted(table)}") from None


def load_block_decoder_from_config(cfg) -> GPTNeoXBlockDecoder:
    cls = _cls(BLOCK_DECODER_CLS, _get(cfg, "block_decoder.cls"), "block_decoder")
    if _get(cfg, "block_decoder.use_pretrained_weights", False):
        raise NotImplementedError("use_pretrained_weights needs hub access; load a local checkpoint instead")
    dt = get_torch_dtype(cfg)
    print("Initializing block decoder from scratch...")
    model = cls(_neox_config(_get(cfg, "block_decoder"), dt),
                n_embedding_tokens=_get(cfg, "embedder.n_embedding_tokens", 1),
                use_block_decoding_loss=_get(cfg, "block_decoding_loss.enable", False),
                block_decoding_loss_type=_get(cfg, "block_decoding_loss.type", "contrastive"),
                keep_unused_embed_in=_get(cfg, "block_decoder.keep_unused_embed_in", True))
    return model.to(dt)


def load_embedder_from_config(cfg, block_decoder) -> LookupEmbedder:
    cls = _cls(EMBEDDER_CLS, _get(cfg, "embedder.cls"), "embedder")
    if _get(cfg, "embedder.use_pretrained_weights", False):
        raise NotImplementedError("use_pretrained_weights needs hub access; load a local checkpoint instead")
    kwargs = {"block_length": _get(cfg, "block_length"),
              "n_embedding_tokens": _get(cfg, "embedder.n_embedding_tokens", 1),
              "projection_method": _get(cfg, "embedder.projection_method"),
              "projection_hidden_size": block_decoder.hidden_size}
    if _get(cfg, "embedder.init_projection") is not None:
        kwargs["init_projection"] = _get(cfg, "embedder.init_projection")
    print("Initializing embedder from scratch...")
    config = LookupConfig()
    for k, v in dict(_get(cfg, "embedder.config", {}) or {}).items():
        if not hasattr(config, k):
            raise ValueError(f"Config key {k} not found in embedder config.")
        setattr(config, k, v)
    return cls(config, **kwargs).to(get_torch_dtype(cfg))


def load_token_decoder_from_config(cfg, block_decoder) -> GPTNeoXTokenDecoder:
    cls = _cls(TOKEN_DECODER_CLS, _get(cfg, "token_decoder.cls"), "token_decoder")
    if _get(cfg, "token_decoder.use_pretrained_weights", False):
        raise NotImplementedError("use_pretrained_weights needs hub access; load a local checkpoint instead")
    kwargs = {"block_length": _get(cfg, "block_length"),
              "n_embedding_tokens": _get(cfg, "embedder.n_embedding_tokens", 1),
              "projection_hidden_size": block_decoder.hidden_size,
              "expansion_method": _get(cfg, "token_decoder.expansion_method"),
              "expansion_ratio": _get(cfg, "token_decoder.expansion_ratio"),
              "decoding_strategy": _get(cfg, "token_decoder.decoding_strategy")}
    if _get(cfg, "token_decoder.init_expansion") is not None:
        kwargs["init_expansion"] = _get(cfg, "token_decoder.init_expansion")
    dt = get_torch_dtype(cfg)
    print("Initializing token decoder from scratch...")
    model = cls(_neox_config(_get(cfg, "token_decoder"), dt), **kwargs).to(dt)
    # build-specific knob (not in the reference YAMLs): RoPE positions of cached generation, see engine.td_position_table.
    # Default "hf439_generate" = the reference's generate(); ``++token_decoder.rope_policy=train`` = plain arange.
    policy = _get(cfg, "token_decoder.rope_policy")
    if policy is not None:
        from ..engine import td_position_table
        td_position_table(policy, 1, 1)   # validates the name
        model.rope_policy = policy
    return model


def _load_tokenizer(cfg):
    path = _get(cfg, "tokenizer.local_path")
    if path is None:
        warnings.warn("no local tokenizer (cfg.tokenizer.local_path): the hub is unreachable, returning tokenizer=None")
        return None
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(path)


def load_block_transformer_from_config(cfg) -> Tuple[BlockTransformer, Optional[Any]]:
    """reference model/utils.py:204-228 (token_mapper is None: both sides use the pythia tokenizer)."""
    if _get(cfg, "tokenizer.embedder", "pythia") != _get(cfg, "tokenizer.token_decoder", "pythia"):
        raise NotImplementedError("different embedder / token-decoder tokenizers need a TokenMapper (out of scope)")
    tokenizer = _load_tokenizer(cfg)
    block_decoder = load_block_decoder_from_config(cfg)
    embedder = load_embedder_from_config(cfg, block_decoder=block_decoder)
    token_decoder = load_token_decoder_from_config(cfg, block_decoder=block_decoder)
    model = BlockTransformer(embedder=embedder, block_decoder=block_decoder, token_decoder=token_decoder,
                             token_mapper=None,
                             use_token_decoding_loss=_get(cfg, "token_decoding_loss.enable", False),
                             use_block_decoding_loss=_get(cfg, "block_decoding_loss.enable", False),
                             block_decoding_loss_weight=_get(cfg, "block_decoding_loss.weight", 1.0),
                             decoding_strategy=_get(cfg, "token_decoder.decoding_strategy"))
    if _get(cfg, "tokenizer.token_decoder", "pythia") == "pythia":
        # differentiate eos and pad: token id 1 is unused by the model (reference model/utils.py:221-226, Q7)
        token_decoder.generation_config.pad_token_id = 1
    return model, tokenizer


def cfg_from_dims(dims, precision: Optional[str] = None) -> dict:
    """A reference-style trainer config dict for a ``block_transformer_amd.config.ModelDims`` (tests, bench, fixtures)."""
    def dec(d):
        return {"cls": "gpt-neo-x", "config": {"num_hidden_layers": d.layers, "hidden_size": d.hidden,
                                               "num_attention_heads": d.heads, "intermediate_size": d.inter,
                                               "rotary_pct": d.rotary_pct, "rotary_emb_base": d.rope_base,
                                               "layer_norm_eps": d.ln_eps, "vocab_size": dims.vocab}}
    return {
        "name": dims.name, "block_length": dims.block_length, "precision": precision or dims.precision,
        "tokenizer": {"embedder": "pythia", "token_decoder": "pythia"},
        "embedder": {"cls": "lookup", "n_embedding_tokens": 1, "projection_method": "concat",
                     "config": {"vocab_size": dims.vocab, "hidden_size": dims.emb_hidden}},
        "block_decoder": dec(dims.bd),
        "token_decoder": dict(dec(dims.td), expansion_method="expansion_layer", expansion_ratio=dims.expansion_ratio,
                              decoding_strategy="prefix"),
        "token_decoding_loss": {"enable": True}, "block_decoding_loss": {"enable": False, "weight": 1.0},
    }


def load_block_transformer_from_dims(dims, precision: Optional[str] = None, keep_unused_embed_in: bool = True,
                                     quiet: bool = True) -> BlockTransformer:
    import contextlib
    import io
    cfg = cfg_from_dims(dims, precision)
    cfg["block_decoder"]["keep_unused_embed_in"] = keep_unused_embed_in
    with (contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, _ = load_block_transformer_from_config(cfg)
    return model
