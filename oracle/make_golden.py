"""Golden-vector generator — runs ONLY in the build container (needs /root/reference); never on the GPU box.

Imports the reference's own classes (LookupEmbedder, GPTNeoXBlockDecoder, GPTNeoXTokenDecoder, BlockTransformer)
through the 3-patch import shim of SURVEY.md Appendix A, loads deterministic random weights
(``oracle.cpu_ref.random_state_dict``; the state-dict key names ARE the reference's), runs the reference's no-cache
paths and writes inputs + expected outputs as small ``tests/golden/<config>.npz`` fixtures.  Nothing from the reference
(source, bytecode) is written anywhere: fixtures are data only.

    python oracle/make_golden.py            # regenerate all fixtures and check the oracle against them

Vectors: G1 embedder, G2 block-decoder prefill hidden states (with and without a key mask),
G4 expansion, G5 token-decoder teacher-forced logits, G6 greedy token ids assembled from the reference classes
through the no-cache path (incl. P % L != 0 and D % L != 0), G7 BlockTransformer.forward logits, G8
preprocess/revert integer cases, G9 = G6 with the token decoder evaluated at the position ids of the reference's cached
generation (prefix rows at 0, token t at t+1: what transformers 4.39.3 ``prepare_inputs_for_generation`` feeds through the
reference's **kwargs; passed here explicitly as ``position_ids=`` to the reference's own ``GPTNeoXTokenDecoder.forward``).
(G3, incremental == prefill row, is a property test in tests/, not a fixture.)
"""
from __future__ import annotations

import hashlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from block_transformer_amd.config import named_dims  # noqa: E402
from oracle.cpu_ref import oracle_from_dims, random_state_dict, td_position_table  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def sd_checksum(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().to(torch.float32).contiguous().numpy().tobytes())
    return h.hexdigest()


def import_reference():
    """SURVEY.md Appendix A shim (transformers 5.15 vs the reference's pinned 4.39.3)."""
    sys.path.insert(0, REF)
    import transformers.models.gpt_neox.modeling_gpt_neox as M
    M.GPTNeoXFlashAttention2 = M.GPTNeoXAttention
    _o = M.GPTNeoXLayer.__init__
    M.GPTNeoXLayer.__init__ = lambda self, config, layer_idx=None: _o(self, config, layer_idx)
    stub = types.ModuleType("model.token_decoder.t5")
    stub.T5TokenDecoder = type("T5TokenDecoder", (), {})
    sys.modules["model.token_decoder.t5"] = stub
    import model.block_decoder.gpt_neo_x as BD
    import model.token_decoder.gpt_neo_x as TD
    from model.embedder.lookup import LookupEmbedder, LookupConfig
    from model.block_transformer import BlockTransformer
    from transformers import GPTNeoXConfig
    return BD, TD, LookupEmbedder, LookupConfig, BlockTransformer, GPTNeoXConfig


def build_reference_model(dims, sd):
    BD, TD, LookupEmbedder, LookupConfig, BlockTransformer, GPTNeoXConfig = import_reference()

    def cfg(dd):
        return GPTNeoXConfig(vocab_size=dims.vocab, hidden_size=dd.hidden, num_hidden_layers=dd.layers,
                             num_attention_heads=dd.heads, intermediate_size=dd.inter, rotary_pct=dd.rotary_pct,
                             max_position_embeddings=2048, bos_token_id=0, eos_token_id=0,
                             use_parallel_residual=True, layer_norm_eps=dd.ln_eps, hidden_act="gelu",
                             attn_implementation="eager")

    bd = BD.GPTNeoXBlockDecoder(cfg(dims.bd), n_embedding_tokens=1)
    td = TD.GPTNeoXTokenDecoder(cfg(dims.td), block_length=dims.L, n_embedding_tokens=1,
                                projection_hidden_size=dims.bd.hidden, expansion_method="expansion_layer",
                                expansion_ratio=dims.E, decoding_strategy="prefix")
    for m in (bd, td):
        for i, layer in enumerate(m.gpt_neox.layers):
            layer.attention.layer_idx = i
    emb = LookupEmbedder(LookupConfig(vocab_size=dims.vocab, hidden_size=dims.emb_hidden), block_length=dims.L,
                         n_embedding_tokens=1, projection_method="concat", projection_hidden_size=dims.bd.hidden)
    model = BlockTransformer(embedder=emb, block_decoder=bd, token_decoder=td, use_token_decoding_loss=True,
                             decoding_strategy="prefix").eval()
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model


@torch.no_grad()
def reference_greedy(model, dims, input_ids, attention_mask, max_length, td_positions=None):
    """Greedy tokens from the reference's own classes through the NO-CACHE path:
    loop {embed all blocks -> block decoder -> last hidden -> L x (token decoder on [BOS, tokens...], argmax)}.
    End test and partial-final-block handling follow model/block_transformer.py:387-388,415."""
    d = model.preprocess_inputs_for_generation(input_ids, attention_mask)
    ids, mask, bmask, pad = d["input_ids"], d["attention_mask"], d["block_attention_mask"], d["initial_block_padding"]
    B, L = ids.shape[0], dims.L
    ctx_trace = []
    while True:
        nb = ids.shape[1]
        emb = model.embed_blocks(ids, mask).view(B, nb, -1)
        hs = model.block_decoder(inputs_embeds=emb, block_attention_mask=bmask, use_cache=False,
                                 return_dict=True).hidden_states
        ctx = hs[:, -1:, :]
        ctx_trace.append(ctx[:, 0].clone())
        count = min(max_length - (nb * L - pad), L)
        toks = torch.zeros(B, 1, dtype=torch.long)  # BOS/EOS placeholder = 0
        for _ in range(count):
            kw = {}
            if td_positions is not None:   # adapted length = E prefix rows + (len(toks) - 1) tokens
                n_adapted = dims.E + toks.shape[1] - 1
                kw["position_ids"] = torch.tensor(td_positions[:n_adapted])[None, :]
            lg = model.token_decoder(input_ids=toks, block_embeddings=ctx, use_cache=False, return_dict=True,
                                     **kw).logits
            toks = torch.cat([toks, lg[:, -1].argmax(-1, keepdim=True)], dim=1)
        new = torch.full((B, L), model.token_decoder.config.pad_token_id, dtype=torch.long)
        new[:, :count] = toks[:, 1:]
        ids = torch.cat([ids, new[:, None]], dim=1)
        mask = torch.cat([mask, torch.ones(B, 1, L, dtype=mask.dtype)], dim=1)
        if ids.shape[1] * L >= max_length:
            break
        bmask = torch.cat([bmask, torch.ones(B, 1, dtype=bmask.dtype)], dim=1)
    out = model.revert_block_input_ids_to_vanilla(ids, pad, L - count)
    return out, torch.stack(ctx_trace, 0)


@torch.no_grad()
def make_fixture(name, seed, store_weights, B=2, nb=5, gen_cases=((10, 23), (8, 16), (7, 12)), bias_std=0.02):
    dims = named_dims(name)
    sd = random_state_dict(dims, seed=seed, bias_std=bias_std)
    model = build_reference_model(dims, sd)
    L, V, H = dims.L, dims.vocab, dims.bd.hidden
    g = torch.Generator().manual_seed(seed + 1000)
    fx = {"seed": np.int64(seed), "bias_std": np.float64(bias_std), "sd_sha256": np.array(sd_checksum(sd))}
    if store_weights:
        for k, v in sd.items():
            fx["w:" + k] = v.numpy()

    # G1 embedder
    ids = torch.randint(0, V, (B, nb, L), generator=g)
    mask = torch.ones_like(ids)
    emb = model.embed_blocks(ids, mask)
    fx["g1_ids"], fx["g1_emb"] = ids.numpy(), emb.reshape(B, nb, H).numpy()

    # G2 block-decoder prefill (no mask / with key mask: first block masked for the last row)
    x = emb.view(B, nb, H)
    bm = torch.ones(B, nb, dtype=torch.long)
    hs = model.block_decoder(inputs_embeds=x, block_attention_mask=bm, use_cache=False, return_dict=True).hidden_states
    fx["g2_hidden"] = hs.numpy()
    bm2 = bm.clone()
    bm2[B - 1, 0] = 0
    hs2 = model.block_decoder(inputs_embeds=x, block_attention_mask=bm2, use_cache=False, return_dict=True).hidden_states
    fx["g2_mask"], fx["g2_hidden_masked"] = bm2.numpy(), hs2.numpy()

    # G4 expansion
    ctx = hs[:, -1:, :]
    fx["g4_ctx"], fx["g4_expanded"] = ctx[:, 0].numpy(), model.token_decoder.expand_block_embeddings(ctx).numpy()

    # G5 token-decoder teacher-forced logits for [BOS, A, B, C...] (L inputs -> L logits rows)
    toks = torch.randint(1, V, (B, L - 1), generator=g)
    tin = torch.cat([torch.zeros(B, 1, dtype=torch.long), toks], dim=1)
    lg = model.token_decoder(input_ids=tin, block_embeddings=ctx, use_cache=False, return_dict=True).logits
    fx["g5_tokens"], fx["g5_logits"] = toks.numpy(), lg.numpy()
    # G9 logits: the same teacher-forced call at the cached-generation position ids (prefix rows 0, token t at t+1)
    pos9 = torch.tensor(td_position_table("hf439_generate", dims.E, L))[None, :]
    lg9 = model.token_decoder(input_ids=tin, block_embeddings=ctx, use_cache=False, return_dict=True,
                              position_ids=pos9).logits
    fx["g9_logits"] = lg9.numpy()
    assert (lg9 - lg).abs().max() > 1e-4, "position ids did not reach the reference's attention layers"

    # G7 BlockTransformer.forward logits [B, nb-1, L, V]
    out = model(input_ids=ids, attention_mask=mask, block_attention_mask=bm)
    fx["g7_logits"] = out["logits"].numpy()

    # G6 greedy tokens (P, max_length) cases: P%L != 0 and D%L != 0 included
    for ci, (P, max_length) in enumerate(gen_cases):
        pid = torch.randint(1, V, (B, P), generator=g)
        toks_out, ctx_trace = reference_greedy(model, dims, pid, None, max_length)
        fx[f"g6_{ci}_prompt"], fx[f"g6_{ci}_max_length"] = pid.numpy(), np.int64(max_length)
        fx[f"g6_{ci}_tokens"], fx[f"g6_{ci}_ctx"] = toks_out.numpy(), ctx_trace.numpy()
        # G9: same prompts, token decoder at the cached-generation position ids
        toks9, ctx9 = reference_greedy(model, dims, pid, None, max_length,
                                       td_positions=td_position_table("hf439_generate", dims.E, L))
        fx[f"g9_{ci}_tokens"], fx[f"g9_{ci}_ctx"] = toks9.numpy(), ctx9.numpy()

    # G8 preprocess / revert integer cases
    for ci, P in enumerate((L * 2, L * 2 + 1, L * 3 - 1)):
        pid = torch.randint(1, V, (B, P), generator=g)
        d = model.preprocess_inputs_for_generation(pid, None)
        fx[f"g8_{ci}_in"] = pid.numpy()
        fx[f"g8_{ci}_ids"], fx[f"g8_{ci}_mask"] = d["input_ids"].numpy(), d["attention_mask"].numpy()
        fx[f"g8_{ci}_bmask"], fx[f"g8_{ci}_pad"] = d["block_attention_mask"].numpy(), np.int64(d["initial_block_padding"])
        rv = model.revert_block_input_ids_to_vanilla(d["input_ids"], d["initial_block_padding"], 0)
        fx[f"g8_{ci}_revert"] = rv.numpy()

    # slim down big-vocab logits: keep strided slices + argmax
    if V > 1024:
        for k in ("g5_logits", "g7_logits", "g9_logits"):
            full = fx.pop(k)
            fx[k + "_argmax"] = full.argmax(-1)
            fx[k + "_slice"] = full[..., ::97].copy()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    np.savez_compressed(path, **fx)
    return path, dims, sd, fx


def check_oracle(dims, sd, fx, tol=2e-5):
    """The restatement must reproduce every vector (fp32, tolerance 2e-5 abs on O(1) activations/logits)."""
    o = oracle_from_dims(dims, sd, rope_policy="train")
    o9 = oracle_from_dims(dims, sd, rope_policy="hf439_generate")
    t = lambda k: torch.from_numpy(np.asarray(fx[k]))
    B, nb, L = t("g1_ids").shape
    H = dims.bd.hidden
    errs = {}
    emb = o.embed(t("g1_ids")).view(B, nb, H)
    errs["g1"] = (emb - t("g1_emb")).abs().max().item()
    hs, _ = o.block_decoder(emb, torch.ones(B, nb, dtype=torch.long))
    errs["g2"] = (hs - t("g2_hidden")).abs().max().item()
    hs2, _ = o.block_decoder(emb, t("g2_mask"))
    # last row / position 0 attends to nothing (fully masked): unspecified, excluded
    d2 = (hs2 - t("g2_hidden_masked")).abs()
    d2[B - 1, 0] = 0
    errs["g2_masked"] = d2.max().item()
    errs["g4"] = (o.expand(t("g4_ctx")) - t("g4_expanded")).abs().max().item()
    lg = o.token_decoder_teacher_forced(t("g4_ctx"), t("g5_tokens"))
    f7 = o.forward_logits(t("g1_ids"), torch.ones_like(t("g1_ids")), torch.ones(B, nb, dtype=torch.long))
    lg9 = o.token_decoder_teacher_forced(t("g4_ctx"), t("g5_tokens"),
                                         positions=td_position_table("hf439_generate", dims.E, L))
    if "g9_logits" in fx:
        errs["g9_logits"] = (lg9 - t("g9_logits")).abs().max().item()
    else:
        errs["g9_logits"] = (lg9[..., ::97] - t("g9_logits_slice")).abs().max().item()
        assert (lg9.argmax(-1) == t("g9_logits_argmax")).all()
    if "g5_logits" in fx:
        errs["g5"] = (lg - t("g5_logits")).abs().max().item()
        errs["g7"] = (f7 - t("g7_logits")).abs().max().item()
    else:
        errs["g5"] = (lg[..., ::97] - t("g5_logits_slice")).abs().max().item()
        errs["g7"] = (f7[..., ::97] - t("g7_logits_slice")).abs().max().item()
        assert (lg.argmax(-1) == t("g5_logits_argmax")).all() and (f7.argmax(-1) == t("g7_logits_argmax")).all()
    ci = 0
    while f"g6_{ci}_prompt" in fx:
        out, trace = o.generate(t(f"g6_{ci}_prompt"), max_length=int(fx[f"g6_{ci}_max_length"]), return_trace=True)
        assert out.shape == t(f"g6_{ci}_tokens").shape, (out.shape, t(f"g6_{ci}_tokens").shape)
        assert (out == t(f"g6_{ci}_tokens")).all(), f"G6 case {ci}: greedy tokens differ"
        errs[f"g6_{ci}_ctx"] = (torch.stack(trace["ctx"], 0) - t(f"g6_{ci}_ctx")).abs().max().item()
        out9, trace9 = o9.generate(t(f"g6_{ci}_prompt"), max_length=int(fx[f"g6_{ci}_max_length"]), return_trace=True)
        assert (out9 == t(f"g9_{ci}_tokens")).all(), f"G9 case {ci}: greedy tokens differ (cached-generation positions)"
        errs[f"g9_{ci}_ctx"] = (torch.stack(trace9["ctx"], 0) - t(f"g9_{ci}_ctx")).abs().max().item()
        ci += 1
    ci = 0
    while f"g8_{ci}_in" in fx:
        d = o.preprocess_inputs_for_generation(t(f"g8_{ci}_in"), None)
        assert (d["input_ids"] == t(f"g8_{ci}_ids")).all() and (d["attention_mask"] == t(f"g8_{ci}_mask")).all()
        assert (d["block_attention_mask"] == t(f"g8_{ci}_bmask")).all()
        assert d["initial_block_padding"] == int(fx[f"g8_{ci}_pad"])
        rv = o.revert_block_input_ids_to_vanilla(d["input_ids"], d["initial_block_padding"], 0)
        assert (rv == t(f"g8_{ci}_revert")).all()
        ci += 1
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, f"oracle deviates from the reference: {bad}"
    return errs


def main():
    specs = [
        ("micro", 11, True, dict()),
        ("micro_l8", 12, False, dict(gen_cases=((10, 29), (16, 32), (9, 20)))),
        ("micro128", 13, False, dict()),
        ("micro64", 14, False, dict()),
        # BASELINE.json configs[0]: block_main_b4 smallest config, batch 1, prefill 16 / decode 16
        ("block_main_b4_5", 15, False, dict(B=1, nb=4, gen_cases=((16, 32),), bias_std=0.0)),
    ]
    for name, seed, store, kw in specs:
        path, dims, sd, fx = make_fixture(name, seed, store, **kw)
        errs = check_oracle(dims, sd, fx)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB  oracle-vs-reference max abs err: "
              + ", ".join(f"{k}={v:.1e}" for k, v in errs.items()))


if __name__ == "__main__":
    main()
