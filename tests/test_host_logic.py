"""CPU-side tests: config rules, format helpers, module surface, C-ABI exports, loud failure without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, t
from block_transformer_amd import _lib
from block_transformer_amd.config import (DecoderDims, block_decoder_max_positions, dims_from_cfg, get_max_length,
                                          head_dim_rule, load_config, named_dims, apply_overrides)
from block_transformer_amd.engine import td_position_table
from block_transformer_amd.model import (BlockTransformer, GPTNeoXBlockDecoder, GPTNeoXConfig, GPTNeoXTokenDecoder,
                                         LookupConfig, LookupEmbedder, load_block_transformer_from_config)
from block_transformer_amd.model.utils import load_block_transformer_from_dims


def test_head_dim_and_intermediate_autofill():
    assert [head_dim_rule(h) for h in (256, 257, 1536, 1537, 2048)] == [32, 64, 64, 128, 128]
    d = DecoderDims.autofill(hidden=2048, layers=12)
    assert (d.heads, d.head_dim, d.inter, d.rot) == (16, 128, 8192, 32)
    d = DecoderDims.autofill(hidden=256, layers=3)
    assert (d.heads, d.head_dim, d.inter, d.rot) == (8, 32, 1024, 8)
    with pytest.raises(ValueError):
        DecoderDims.autofill(hidden=100, layers=1)


def test_named_configs():
    d = named_dims("block_main_b4_1.2b")
    assert (d.L, d.E, d.vocab, d.emb_hidden, d.bd.hidden, d.bd.layers, d.local_len) == (4, 2, 50304, 512, 2048, 12, 5)
    assert d.kv_bytes_per_position() == 98304                      # kv_pos
    assert d.bd.param_count() * 2 == 1208606720                    # W_dec = 1.2086 GB
    b8 = named_dims("block_main_b8_1.2b")
    assert (b8.L, b8.emb_hidden, b8.local_len) == (8, 256, 9)
    with pytest.raises(ValueError):
        dims_from_cfg({"block_length": 4, "embedder": {"cls": "lookup", "config": {"hidden_size": 100}},
                       "block_decoder": {"config": {"hidden_size": 256, "num_hidden_layers": 1}},
                       "token_decoder": {"expansion_ratio": 2, "config": {"hidden_size": 256, "num_hidden_layers": 1}}})


def test_max_length_formulas():
    assert get_max_length(4, 2048, 128) == 2176
    assert get_max_length(4, 2047, 128) == 2176
    assert get_max_length(None, 10, 5) == 15
    assert block_decoder_max_positions(4, 2048, 128) == 543
    assert block_decoder_max_positions(4, 128, 2048) == 543


def test_overrides():
    cfg = apply_overrides(load_config("block_main_b4_5"), ["++block_decoder.config.num_hidden_layers=2", "precision=fp32"])
    assert cfg["block_decoder"]["config"]["num_hidden_layers"] == 2 and cfg["precision"] == "fp32"


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports exactly what include/bt_hip.h declares (no compute without a GPU)."""
    lib = _lib.lib()
    assert lib.bt_version() == 310
    header = open(os.path.join(ROOT, "include", "bt_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int32_t|int64_t|const char\*)\s+(bt_\w+)\(", header, flags=re.M))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    # argument validation happens before any launch: usable without a GPU
    rc = lib.bt_linear(7, 1, 1, 1, None, 1, None, None, None, 0, None, 1, 0, None)
    assert rc == -1 and b"dtype" in lib.bt_last_error()
    rc = lib.bt_attention(1, 0, 1, 1, 1, 48, 1, 48, 1, 1, 8, None, 0, None, 0, 1, 48, 1, None, 0, None)
    assert rc == -1 and b"head_dim" in lib.bt_last_error()
    rc = lib.bt_block_step_enqueue(None, 4, 1, None)
    assert rc == -1 and b"descriptor" in lib.bt_last_error()
    rc = lib.bt_topk_sample(1, 1, 50304, 1, 50304, 1.0, 2000, 0, None, 0, 1, 1, None)   # top_k beyond the candidate buffer
    assert rc == -2 and b"top_k" in lib.bt_last_error()
    # bt_compact_enqueue: dst must describe the same KV storage as src and own its state buffers (checked before any launch)
    import ctypes
    src, dst = _lib.bt_engine_desc(), _lib.bt_engine_desc()
    src.B, dst.B, src.gk, src.gv, dst.gk, dst.gv = 8, 4, 0x1000, 0x2000, 0x1000, 0x3000
    rc = lib.bt_compact_enqueue(ctypes.byref(src), ctypes.byref(dst), 0x10, 2, None)
    assert rc == -1 and b"same global KV storage" in lib.bt_last_error()
    dst.gv = 0x2000
    rc = lib.bt_compact_enqueue(ctypes.byref(src), ctypes.byref(dst), 0x10, 5, None)
    assert rc == -1 and b"row counts" in lib.bt_last_error()
    rc = lib.bt_compact_enqueue(ctypes.byref(src), ctypes.byref(dst), 0x10, 2, None)
    assert rc == -1 and b"null state buffer" in lib.bt_last_error()
    # struct_size guard: a descriptor / attention-args struct of another header version is refused, not read past its end
    bad = _lib.bt_engine_desc()
    bad.struct_size = ctypes.sizeof(_lib.bt_engine_desc) - 8
    rc = lib.bt_block_step_enqueue(ctypes.byref(bad), 4, 1, None)
    assert rc == -1 and b"struct_size" in lib.bt_last_error()
    rc = lib.bt_compact_enqueue(ctypes.byref(bad), ctypes.byref(dst), 0x10, 2, None)
    assert rc == -1 and b"struct_size" in lib.bt_last_error()
    args = _lib.bt_attn_args()
    assert args.struct_size == ctypes.sizeof(_lib.bt_attn_args) and _lib.bt_engine_desc().struct_size == ctypes.sizeof(_lib.bt_engine_desc)
    args.struct_size = 0
    dims, w = _lib.bt_decoder_dims(64, 2, 32, 8, 256, 1, 1e-5), _lib.bt_decoder_weights()
    lw = _lib.bt_layer_weights()
    w.layers_host = ctypes.pointer(lw)
    rc = lib.bt_decoder_forward(1, ctypes.byref(dims), ctypes.byref(w), 1, 1, 0x1000, ctypes.byref(args), 0, 0x1000, 0x1000, 1 << 20, None)
    assert rc == -1 and b"struct_size" in lib.bt_last_error()
    # split-N rule shared by the library and the engine's workspace sizing
    assert lib.bt_attention_split_count(128, 544, 1) == 1 and lib.bt_attention_split_count(128, 544, 4) == 4
    assert lib.bt_attention_split_count(128, 40000, 1) == 3   # 40000 keys do not fit one 64 KiB score row


def test_product_library_has_no_debug_hooks():
    """libbt_hip.so (the product) exports no bt_debug_* setter and contains none of the A/B or timing-ablation kernels; the
    hooks live in the separate test build libbt_hip_dbg.so (-DBT_DEBUG_HOOKS)."""
    import subprocess
    prod = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "bt_linear" in prod and "bt_debug" not in prod
    # the symbol table IS the C ABI: every dynamic symbol the product defines is a bt_* function of include/bt_hip.h --
    # no C++-mangled internals (namespace bt), no kernel host stubs (csrc/exports.map)
    names = [ln.split()[-1] for ln in prod.splitlines() if ln.strip()]
    assert names and sorted(names) == sorted(_lib.EXPORTED_SYMBOLS), sorted(set(names) ^ set(_lib.EXPORTED_SYMBOLS))
    blob = open(_lib.LIB_PATH, "rb").read()
    for kern in (b"gemm_bf16_mfma256_kernelILb", b"flash_prefill_kernelILi", b"tr_probe_kernel", b"Mma32"):
        assert kern not in blob, kern
    dbg = subprocess.run(["nm", "-D", "--defined-only", _lib.DEBUG_LIB_PATH], capture_output=True, text=True).stdout
    for name in _lib._DEBUG_SIGS:
        assert name in dbg, name
    assert all(ln.split()[-1].startswith("bt_") for ln in dbg.splitlines() if ln.strip())
    prev = _lib.use_debug_library(True)
    try:
        assert _lib.lib().bt_version() == 310 and hasattr(_lib.lib(), "bt_debug_gemm_variant")
    finally:
        _lib.use_debug_library(prev)
    assert not hasattr(_lib.lib(), "bt_debug_gemm_variant_missing")


def test_engine_desc_layout_matches_header(tmp_path):
    """ctypes bt_engine_desc / bt_attn_args == the C structs of include/bt_hip.h (size and every field offset), checked
    against a probe compiled from the header with gcc."""
    import ctypes
    import subprocess
    structs = {"bt_engine_desc": _lib.bt_engine_desc, "bt_attn_args": _lib.bt_attn_args,
               "bt_decoder_dims": _lib.bt_decoder_dims, "bt_layer_weights": _lib.bt_layer_weights,
               "bt_decoder_weights": _lib.bt_decoder_weights}
    lines = ['#include "bt_hip.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {"]
    for name, cls in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f, _ in cls._fields_:
            lines.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    lines += ["return 0; }"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(out[name]) == ctypes.sizeof(cls), name
        for f, _ in cls._fields_:
            assert int(out[f"{name}.{f}"]) == getattr(cls, f).offset, f"{name}.{f}"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_handles", {})
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no fallback"):
        _lib.lib()


def test_no_cpu_fallback():
    model = load_block_transformer_from_dims(named_dims("micro"), "fp32")
    ids = torch.randint(1, 200, (2, 8))
    with pytest.raises(RuntimeError, match="GPU only"):
        model.generate(ids, max_length=16)
    with pytest.raises(_lib.BtError, match="GPU memory only"):
        model.embedder(ids.view(-1, 4))
    with pytest.raises(_lib.BtError):
        model.block_decoder(inputs_embeds=torch.zeros(1, 2, 64), block_attention_mask=torch.ones(1, 2, dtype=torch.long))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "block_transformer_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} references the oracle"


def test_state_dict_names_match_reference(golden):
    dims, sd, fx = golden("micro")
    model = load_block_transformer_from_dims(dims, "fp32")
    ref_keys = sorted(k[2:] for k in fx if k.startswith("w:"))
    assert sorted(model.state_dict().keys()) == ref_keys
    model.load_state_dict(sd, strict=True)
    assert model.block_length == 4 and model.n_embedding_tokens == 1 and model.projection_hidden_size == 64
    assert model.config is model.token_decoder.config and model.config.vocab_size == 256
    assert model.token_decoder.n_expanded_emb == 2 and model.block_decoder.hidden_size == 64
    assert model.token_decoder.config.pad_token_id == model.token_decoder.config.eos_token_id == 0
    assert model.token_decoder.generation_config.pad_token_id == 1   # factory, Q7
    unused = model.token_decoder.generation_config.update(do_sample=True, temperature=0.7, bogus=1)
    assert unused == {"bogus": 1} and model.token_decoder.generation_config.do_sample


def test_preprocess_and_revert_match_reference_vectors(golden):
    for name in ("micro", "micro_l8"):
        dims, sd, fx = golden(name)
        model = load_block_transformer_from_dims(dims, "fp32")
        ci = 0
        while f"g8_{ci}_in" in fx:
            d = model.preprocess_inputs_for_generation(t(fx, f"g8_{ci}_in"))
            assert torch.equal(d["input_ids"], t(fx, f"g8_{ci}_ids"))
            assert torch.equal(d["attention_mask"], t(fx, f"g8_{ci}_mask"))
            assert torch.equal(d["block_attention_mask"], t(fx, f"g8_{ci}_bmask"))
            assert d["initial_block_padding"] == int(fx[f"g8_{ci}_pad"])
            rv = BlockTransformer.revert_block_input_ids_to_vanilla(d["input_ids"], d["initial_block_padding"], 0)
            assert torch.equal(rv, t(fx, f"g8_{ci}_revert"))
            ci += 1
    # last_block_unfilled_length strips the tail
    x = torch.arange(24).view(1, 6, 4)
    assert BlockTransformer.revert_block_input_ids_to_vanilla(x, 1, 2).tolist() == [list(range(1, 22))]


def test_generate_early_return_q1():
    model = load_block_transformer_from_dims(named_dims("micro"), "fp32")
    ids = torch.randint(1, 200, (2, 8))
    with pytest.warns(UserWarning, match="max_length is less"):
        out = model.generate(ids, max_length=4)
    assert torch.equal(out, ids)


def test_constructor_contracts():
    cfg = GPTNeoXConfig(vocab_size=64, hidden_size=64, num_hidden_layers=1, num_attention_heads=2)
    with pytest.raises(NotImplementedError):
        GPTNeoXBlockDecoder(cfg, n_embedding_tokens=2)
    with pytest.raises(ValueError, match="expansion_ratio"):
        GPTNeoXTokenDecoder(cfg, block_length=4, n_embedding_tokens=1, projection_hidden_size=64,
                            expansion_method="expansion_layer", expansion_ratio=None, decoding_strategy="prefix")
    with pytest.raises(NotImplementedError):
        GPTNeoXTokenDecoder(cfg, block_length=4, n_embedding_tokens=1, projection_hidden_size=64,
                            expansion_method="expansion_layer", expansion_ratio=4, decoding_strategy="summation")
    with pytest.raises(ValueError, match="config.hidden_size must be equal"):
        LookupEmbedder(LookupConfig(vocab_size=64, hidden_size=8), block_length=4, n_embedding_tokens=1,
                       projection_method="concat", projection_hidden_size=64)
    with pytest.raises(ValueError, match="vocab_size"):
        LookupEmbedder(LookupConfig(hidden_size=16), block_length=4, n_embedding_tokens=1, projection_hidden_size=64)
    emb = LookupEmbedder(LookupConfig(vocab_size=64), block_length=4, n_embedding_tokens=1, projection_hidden_size=64)
    assert emb.config.hidden_size == 16 and emb.embeddings.weight.shape == (64, 16)
    cfg2 = load_config("block_main_b4_5")
    cfg2["token_decoder"]["cls"] = "t5"
    with pytest.raises(NotImplementedError, match="outside the MI355X decode path"):
        load_block_transformer_from_config(cfg2)


def test_td_position_table():
    assert td_position_table("train", 2, 8) == list(range(9))
    assert td_position_table("hf439_generate", 2, 8) == [0, 0, 1, 2, 3, 4, 5, 6, 7]


def test_batch_tuner_bookkeeping():
    """The reference's batch auto-tune is pure arithmetic over measurement records (measure_generation_time.py:250-372)."""
    from block_transformer_amd import measure_generation_time as H
    assert H.get_number_of_repetitions(255) == 10 and H.get_number_of_repetitions(256) == 5
    assert [H.round_batch(v) for v in (31, 33, 130, 600, 3000)] == [31, 32, 128, 576, 3072]
    rec = lambda b, m=None, oom=False: {"prefill_length": 2048, "decode_length": 128, "batch_size": b, "oom": oom,  # noqa: E731
                                        "max_memory_allocated": m}
    data = [rec(1, 3.05), rec(2, 3.10)]
    assert H.predict_memory_per_sample(data, 2048, 128) == pytest.approx(0.05)
    assert H.predict_memory_per_sample(data[:1], 2048, 128) is None
    # no OOM yet: aim halfway to the predicted 100 % VRAM batch: b2 = 2 + int((288-3.1)/0.05) = 5700 -> middle 2851 -> 2816
    assert H.find_next_batch_size(data, 2048, 128, 288.0) == 2816
    data.append(rec(2816, 143.9))
    nxt = H.find_next_batch_size(data, 2048, 128, 288.0)
    assert nxt is not None and nxt > 2816
    data.append(rec(nxt, None, oom=True))
    assert H.predict_oom(data, nxt + 1, 2048, 128) and not H.predict_oom(data, nxt - 1, 2048, 128)
    mid = H.find_next_batch_size(data, 2048, 128, 288.0)
    assert 2816 < mid < nxt
    assert H.measurement_exists(data, 2816, 2048, 128) and not H.measurement_exists(data, 2816, 128, 2048)
    assert H.get_max_batch_size_row(data, 2048, 128)["batch_size"] == 2816
    assert H.get_max_batch_size_row(data, 2048, 128, oom=True)["batch_size"] == nxt
    # degenerate slope at B=1,2 -> 0.1 GiB, like the reference
    assert H.predict_memory_per_sample([rec(1, 3.0), rec(2, 3.0)], 2048, 128) == 0.1
    # search ends when the bracket closes
    closed = [rec(1, 1.0), rec(2, 2.0), rec(3, None, oom=True)]
    assert H.find_next_batch_size(closed, 2048, 128, 288.0) is None
