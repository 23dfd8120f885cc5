"""GPT-NeoX parameter containers with the HF module tree / state-dict names, and a minimal config object.

The reference builds its decoders from HuggingFace ``GPTNeoXModel`` (transformers==4.39.3).  Here the modules only HOLD
parameters under the same attribute paths (``gpt_neox.layers.N.attention.query_key_value.weight`` ...), so reference
checkpoints load unchanged (checkpoint row); the arithmetic runs in libbt_hip.so on those tensors in
their HF layout."""
from __future__ import annotations

import copy
from typing import Optional

import torch
from torch import nn


class GPTNeoXConfig:
    """The subset of HF GPTNeoXConfig the decode path reads (defaults = pythia-410m-deduped base values, which the
    reference pulls through AutoConfig.from_pretrained, model/utils.py:131-161)."""

    model_type = "gpt_neox"

    def __init__(self, vocab_size=50304, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                 intermediate_size=None, hidden_act="gelu", rotary_pct=0.25, rotary_emb_base=10000,
                 max_position_embeddings=2048, initializer_range=0.02, layer_norm_eps=1e-5, use_cache=True,
                 bos_token_id=0, eos_token_id=0, pad_token_id=None, tie_word_embeddings=False,
                 use_parallel_residual=True, attn_implementation=None, torch_dtype=None, **kwargs):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size if intermediate_size is not None else 4 * hidden_size
        self.hidden_act = hidden_act
        self.rotary_pct = rotary_pct
        self.rotary_emb_base = rotary_emb_base
        self.max_position_embeddings = max_position_embeddings
        self.initializer_range = initializer_range
        self.layer_norm_eps = layer_norm_eps
        self.use_cache = use_cache
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.pad_token_id = pad_token_id
        self.tie_word_embeddings = tie_word_embeddings
        self.use_parallel_residual = use_parallel_residual
        self._attn_implementation = attn_implementation
        self.torch_dtype = torch_dtype
        self.use_return_dict = True
        for k, v in kwargs.items():
            setattr(self, k, v)
        if hidden_size % num_attention_heads != 0:
            raise ValueError("The hidden size is not divisble by the number of attention heads! Make sure to update them!")
        if not use_parallel_residual:
            raise NotImplementedError("only use_parallel_residual=True (all block_main_* configs) is implemented")
        if hidden_act != "gelu":
            raise NotImplementedError("only hidden_act='gelu' (exact erf GELU) is implemented")

    def to_dict(self):
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_")}

    def copy(self):
        return copy.deepcopy(self)


class GenerationConfig:
    """The knobs the reference touches on ``token_decoder.generation_config`` (inference_demo.py:62-66,
    measure_generation_time.py:90, model/utils.py:221-226)."""

    def __init__(self, eos_token_id=None, pad_token_id=None, bos_token_id=None, do_sample=False, temperature=1.0,
                 top_k=50, max_length=20):
        self.eos_token_id = eos_token_id
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.do_sample = do_sample
        self.temperature = temperature
        self.top_k = top_k
        self.max_length = max_length

    def update(self, **kwargs):
        unused = {}
        for k, v in kwargs.items():
            if hasattr(self, k):
                setattr(self, k, v)
            else:
                unused[k] = v
        return unused


class _Attention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.query_key_value = nn.Linear(config.hidden_size, 3 * config.hidden_size, bias=True)
        self.dense = nn.Linear(config.hidden_size, config.hidden_size, bias=True)


class _MLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense_h_to_4h = nn.Linear(config.hidden_size, config.intermediate_size)
        self.dense_4h_to_h = nn.Linear(config.intermediate_size, config.hidden_size)


class GPTNeoXLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.input_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.post_attention_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.attention = _Attention(config)
        self.mlp = _MLP(config)


class GPTNeoXModel(nn.Module):
    """Parameter container: embed_in, layers[...], final_layer_norm."""

    def __init__(self, config, with_embed_in: bool = True):
        super().__init__()
        self.config = config
        self.embed_in = nn.Embedding(config.vocab_size, config.hidden_size) if with_embed_in else None
        self.layers = nn.ModuleList([GPTNeoXLayer(config) for _ in range(config.num_hidden_layers)])
        self.final_layer_norm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, *a, **k):
        raise RuntimeError("GPTNeoXModel here only holds parameters; use the owning decoder's forward()")


def init_neox_weights(module: nn.Module, initializer_range: float = 0.02) -> None:
    """HF GPTNeoXPreTrainedModel._init_weights: N(0, initializer_range) Linear/Embedding weights, zero biases, LN (1,0)."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            m.weight.data.normal_(mean=0.0, std=initializer_range)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.Embedding):
            m.weight.data.normal_(mean=0.0, std=initializer_range)
        elif isinstance(m, nn.LayerNorm):
            m.bias.data.zero_()
            m.weight.data.fill_(1.0)


class ModelOutput(dict):
    """Attribute + item + index access, like HF's CausalLMOutputWithPast."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in self.values() if v is not None][k]
        return super().__getitem__(k)


def CausalLMOutputWithPast(loss=None, logits=None, past_key_values=None, hidden_states=None, attentions=None):
    return ModelOutput(loss=loss, logits=logits, past_key_values=past_key_values, hidden_states=hidden_states,
                       attentions=attentions)
