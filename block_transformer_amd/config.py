"""Model dimensions and the YAML config reader for the Block Transformer decode path.

Mirrors the parts of the reference's config handling the decode path depends on:

* ``preprocess_config`` autofill rules (reference ``util/config.py:86-105``): head dimension chosen from
  ``hidden_size`` (<=256 -> 32, <=1536 -> 64, else 128) and ``intermediate_size = 4 * hidden_size``.
* the pythia-410m-deduped base values the reference pulls through ``AutoConfig.from_pretrained``
  (``model/utils.py:131-161``): rotary_pct 0.25, rotary base 10000, layer_norm_eps 1e-5, erf GELU,
  parallel residual, untied embeddings, bos = eos = 0.
* ``measure_generation_time.py:44-70``: ``get_max_length`` and the block decoder's ``max_position_embeddings``.

Hydra/OmegaConf are not needed (and not installed): the YAML files are read with PyYAML and only the keys the
decode path consumes are interpreted (``block_length``, ``embedder.config``, ``block_decoder.config``,
``token_decoder.{config,expansion_ratio,expansion_method,decoding_strategy}``, ``precision``).
"""
from __future__ import annotations

import dataclasses
import math
import os
from typing import Any, Dict, Optional

import yaml

CONF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf")

# pythia-410m-deduped base config values (what AutoConfig.from_pretrained would deliver to the reference).
PYTHIA_BASE = dict(
    rotary_pct=0.25,
    rotary_emb_base=10000.0,
    layer_norm_eps=1e-5,
    hidden_act="gelu",
    use_parallel_residual=True,
    bos_token_id=0,
    eos_token_id=0,
    vocab_size=50304,
    max_position_embeddings=2048,
    initializer_range=0.02,
    tie_word_embeddings=False,
)


def head_dim_rule(hidden_size: int) -> int:
    """reference util/config.py:92-98."""
    if hidden_size <= 256:
        return 32
    if hidden_size <= 1536:
        return 64
    return 128


@dataclasses.dataclass(frozen=True)
class DecoderDims:
    """Shape of one GPT-NeoX stack (block decoder or token decoder)."""

    hidden: int
    heads: int
    layers: int
    inter: int
    rotary_pct: float = 0.25
    rope_base: float = 10000.0
    ln_eps: float = 1e-5

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def rot(self) -> int:
        # HF GPTNeoXAttention: rotary_ndims = int(head_size * rotary_pct)
        return int(self.head_dim * self.rotary_pct)

    def param_count(self) -> int:
        h, i = self.hidden, self.inter
        per_layer = 4 * h + (3 * h * h + 3 * h) + (h * h + h) + (h * i + i) + (i * h + h)
        return self.layers * per_layer + 2 * h

    @staticmethod
    def autofill(hidden: int, layers: int, heads: Optional[int] = None, inter: Optional[int] = None,
                 **kw: Any) -> "DecoderDims":
        if heads is None:
            hd = head_dim_rule(hidden)
            if hidden % hd != 0:
                raise ValueError(f"hidden_size must be divisible by {hd}")
            heads = hidden // hd
        if inter is None:
            inter = 4 * hidden
        return DecoderDims(hidden=hidden, heads=heads, layers=layers, inter=inter, **kw)


@dataclasses.dataclass(frozen=True)
class ModelDims:
    """Everything the decode engine needs to size its buffers."""

    name: str
    vocab: int
    block_length: int          # L
    emb_hidden: int            # h_e = H_bd / L  (concat projection, reference model/embedder/base.py:45-61)
    expansion_ratio: int       # E (n_embedding_tokens == 1, so n_expanded_emb == E)
    bd: DecoderDims
    td: DecoderDims
    precision: str = "bf16"
    bos_token_id: int = 0
    eos_token_id: int = 0

    def __post_init__(self):
        if self.emb_hidden * self.block_length != self.bd.hidden:
            raise ValueError(
                "projection_hidden_size must be equal to config.hidden_size * (block_length // n_embedding_tokens), "
                f"but got {self.bd.hidden} and {self.emb_hidden * self.block_length}")

    @property
    def L(self) -> int:
        return self.block_length

    @property
    def E(self) -> int:
        return self.expansion_ratio

    @property
    def local_len(self) -> int:
        """Maximum number of positions the token decoder ever attends over: E prefix rows + L-1 tokens."""
        return self.expansion_ratio + self.block_length - 1

    def kv_bytes_per_position(self, itemsize: int = 2) -> int:
        return self.bd.layers * 2 * self.bd.hidden * itemsize


def get_max_length(block_length: Optional[int], prefill_length: int, decode_length: int) -> int:
    """reference measure_generation_time.py:44-49."""
    if block_length:
        prefill_blocks = math.ceil(prefill_length / block_length)
        return prefill_blocks * block_length + decode_length
    return prefill_length + decode_length


def block_decoder_max_positions(block_length: int, prefill_length: int, decode_length: int) -> int:
    """reference measure_generation_time.py:65-70: max_position_embeddings = ceil(max_length / L) - 1."""
    max_length = get_max_length(block_length, prefill_length, decode_length)
    return math.ceil(max_length / block_length) - 1


def _decoder_from_cfg(node: Dict[str, Any]) -> DecoderDims:
    c = dict(node.get("config") or {})
    return DecoderDims.autofill(
        hidden=int(c["hidden_size"]),
        layers=int(c["num_hidden_layers"]),
        heads=c.get("num_attention_heads"),
        inter=c.get("intermediate_size"),
        rotary_pct=float(c.get("rotary_pct", PYTHIA_BASE["rotary_pct"])),
        rope_base=float(c.get("rotary_emb_base", PYTHIA_BASE["rotary_emb_base"])),
        ln_eps=float(c.get("layer_norm_eps", PYTHIA_BASE["layer_norm_eps"])),
    )


def dims_from_cfg(cfg: Dict[str, Any]) -> ModelDims:
    """Interpret a reference-style trainer YAML dict (conf/trainer/block_main_*.yaml keys)."""
    block_length = cfg.get("block_length")
    if block_length is None:
        split = cfg.get("block_split") or {}
        block_length = (split.get("distribution_kwargs") or {}).get("length")
    if block_length is None:
        raise ValueError("Config does not include block_length but block mode is expected.")
    if cfg.get("precision", "bf16") not in ("fp32", "fp16", "bf16"):
        raise NotImplementedError(f"Precision {cfg.get('precision')} is not implemented yet")
    emb = cfg["embedder"]
    if emb.get("cls", "lookup") != "lookup":
        raise NotImplementedError("only the lookup embedder is on the decode hot path (SURVEY.md section 8)")
    if int(emb.get("n_embedding_tokens", 1)) != 1:
        raise NotImplementedError("generation currently only implemented for n_embedding_tokens == 1")
    td = cfg["token_decoder"]
    if td.get("decoding_strategy", "prefix") != "prefix":
        raise NotImplementedError("only decoding_strategy='prefix' is on the decode hot path")
    if td.get("expansion_method", "expansion_layer") != "expansion_layer":
        raise NotImplementedError("only expansion_method='expansion_layer' is on the decode hot path")
    if td.get("expansion_ratio") is None:
        raise ValueError("`expansion_ratio` must be specified for decoding_strategy='prefix'")
    bd_dims = _decoder_from_cfg(cfg["block_decoder"])
    td_dims = _decoder_from_cfg(td)
    emb_cfg = emb.get("config") or {}
    vocab = int(emb_cfg.get("vocab_size", PYTHIA_BASE["vocab_size"]))
    emb_hidden = emb_cfg.get("hidden_size")
    if emb_hidden is None:
        emb_hidden = bd_dims.hidden // int(block_length)
    return ModelDims(
        name=str(cfg.get("name", "unnamed")),
        vocab=vocab,
        block_length=int(block_length),
        emb_hidden=int(emb_hidden),
        expansion_ratio=int(td["expansion_ratio"]),
        bd=bd_dims,
        td=td_dims,
        precision=str(cfg.get("precision", "bf16")),
    )


def load_config(config_path: str) -> Dict[str, Any]:
    """Counterpart of reference util/config.py:10-13 (returns a plain dict instead of a DictConfig)."""
    if not os.path.isabs(config_path) and not os.path.exists(config_path):
        cand = os.path.join(CONF_DIR, config_path)
        if not cand.endswith(".yaml"):
            cand += ".yaml"
        config_path = cand
    with open(config_path) as f:
        cfg = yaml.safe_load(f)
    return cfg


def apply_overrides(cfg: Dict[str, Any], overrides) -> Dict[str, Any]:
    """Hydra-style ``++a.b.c=value`` / ``a.b=value`` overrides (reference README.md:133-138 usage)."""
    for ov in overrides or []:
        key, _, val = ov.lstrip("+").partition("=")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    return cfg


def named_dims(name: str) -> ModelDims:
    """Named shapes used by tests, bench and fixtures."""
    if name in _SYNTHETIC:
        return _SYNTHETIC[name]()
    return dataclasses.replace(dims_from_cfg(load_config(name)), name=name)


def _micro(L: int = 4, name: str = "micro") -> ModelDims:
    d = DecoderDims(hidden=64, heads=2, layers=2, inter=256)
    return ModelDims(name=name, vocab=256, block_length=L, emb_hidden=64 // L, expansion_ratio=2, bd=d, td=d,
                     precision="fp32")


def _micro128() -> ModelDims:
    d = DecoderDims(hidden=256, heads=2, layers=1, inter=1024)
    return ModelDims(name="micro128", vocab=512, block_length=4, emb_hidden=64, expansion_ratio=2, bd=d, td=d,
                     precision="fp32")


def _micro64() -> ModelDims:
    # head_dim 64 (the 300M/800M-class configs), distinct block/token decoder depths
    bd = DecoderDims(hidden=128, heads=2, layers=2, inter=512)
    td = DecoderDims(hidden=128, heads=2, layers=1, inter=512)
    return ModelDims(name="micro64", vocab=320, block_length=4, emb_hidden=32, expansion_ratio=2, bd=bd, td=td,
                     precision="fp32")


def _b8() -> ModelDims:
    """Synthetic L=8 variant of block_main_b4_1.2b (BASELINE.json configs[4]; SURVEY.md headline finding 4)."""
    base = dims_from_cfg(load_config("block_main_b4_1.2b"))
    return dataclasses.replace(base, name="block_main_b8_1.2b", block_length=8, emb_hidden=base.bd.hidden // 8)


_SYNTHETIC = {
    "micro": lambda: _micro(4, "micro"),
    "micro_l8": lambda: _micro(8, "micro_l8"),
    "micro128": _micro128,
    "micro64": _micro64,
    "block_main_b8_1.2b": _b8,
}
