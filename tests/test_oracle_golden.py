"""The CPU oracle must reproduce every golden vector captured from the reference's own classes (oracle/make_golden.py).
fp32, tolerance 2e-5 absolute on O(1) activations / logits; greedy token ids bit-exact."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN_CONFIGS, t
from oracle.cpu_ref import oracle_from_dims, td_position_table

TOL = 2e-5


def _sha(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().to(torch.float32).contiguous().numpy().tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
def test_weights_regenerate_identically(golden, name):
    dims, sd, fx = golden(name)
    assert _sha(sd) == str(fx["sd_sha256"]), "seeded weights differ from the ones the fixture was generated with"


def test_micro_stored_weights_match_seeded(golden):
    dims, sd, fx = golden("micro")
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), fx["w:" + k]), k


@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
def test_oracle_matches_reference_vectors(golden, name):
    dims, sd, fx = golden(name)
    o = oracle_from_dims(dims, sd)
    B, nb, L = fx["g1_ids"].shape
    H = dims.bd.hidden
    emb = o.embed(t(fx, "g1_ids")).view(B, nb, H)
    assert torch.equal(emb, t(fx, "g1_emb"))                                          # G1 (gather: bit-exact)
    ones = torch.ones(B, nb, dtype=torch.long)
    hs, _ = o.block_decoder(emb, ones)
    assert (hs - t(fx, "g2_hidden")).abs().max() <= TOL                               # G2
    hs2, _ = o.block_decoder(emb, t(fx, "g2_mask"))
    d2 = (hs2 - t(fx, "g2_hidden_masked")).abs()
    d2[B - 1, 0] = 0  # that query sees no key at all: unspecified in the reference (finfo.min arithmetic)
    assert d2.max() <= TOL
    assert (o.expand(t(fx, "g4_ctx")) - t(fx, "g4_expanded")).abs().max() <= TOL      # G4
    lg = o.token_decoder_teacher_forced(t(fx, "g4_ctx"), t(fx, "g5_tokens"))           # G5 (a direct forward: arange)
    lg9 = o.token_decoder_teacher_forced(t(fx, "g4_ctx"), t(fx, "g5_tokens"),          # G9 (cached-generation position ids)
                                         positions=td_position_table("hf439_generate", dims.E, dims.L))
    if "g9_logits" in fx:
        assert (lg9 - t(fx, "g9_logits")).abs().max() <= TOL
        assert (lg9 - t(fx, "g5_logits")).abs().max() > 10 * TOL, "G9 must discriminate the two position policies"
    else:
        assert (lg9[..., ::97] - t(fx, "g9_logits_slice")).abs().max() <= TOL
        assert torch.equal(lg9.argmax(-1), t(fx, "g9_logits_argmax"))
    f7 = o.forward_logits(t(fx, "g1_ids"), torch.ones_like(t(fx, "g1_ids")), ones)     # G7
    if "g5_logits" in fx:
        assert (lg - t(fx, "g5_logits")).abs().max() <= TOL
        assert (f7 - t(fx, "g7_logits")).abs().max() <= TOL
    else:
        assert (lg[..., ::97] - t(fx, "g5_logits_slice")).abs().max() <= TOL
        assert (f7[..., ::97] - t(fx, "g7_logits_slice")).abs().max() <= TOL
        assert torch.equal(lg.argmax(-1), t(fx, "g5_logits_argmax"))
        assert torch.equal(f7.argmax(-1), t(fx, "g7_logits_argmax"))


@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
def test_oracle_greedy_tokens_match_reference(golden, name):                           # G6 (arange) and G9 (generate())
    dims, sd, fx = golden(name)
    assert oracle_from_dims(dims, sd).rope_policy == "hf439_generate", "the oracle's generate() mirrors the reference's"
    for policy, key in (("train", "g6"), ("hf439_generate", "g9")):
        o = oracle_from_dims(dims, sd, rope_policy=policy)
        ci = 0
        while f"g6_{ci}_prompt" in fx:
            out, trace = o.generate(t(fx, f"g6_{ci}_prompt"), max_length=int(fx[f"g6_{ci}_max_length"]), return_trace=True)
            assert torch.equal(out, t(fx, f"{key}_{ci}_tokens")), f"{key} case {ci}"
            assert (torch.stack(trace["ctx"], 0) - t(fx, f"{key}_{ci}_ctx")).abs().max() <= TOL
            ci += 1
        assert ci >= 1


@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
def test_oracle_preprocess_revert(golden, name):                                       # G8
    dims, sd, fx = golden(name)
    o = oracle_from_dims(dims, sd)
    ci = 0
    while f"g8_{ci}_in" in fx:
        d = o.preprocess_inputs_for_generation(t(fx, f"g8_{ci}_in"), None)
        assert torch.equal(d["input_ids"], t(fx, f"g8_{ci}_ids"))
        assert torch.equal(d["attention_mask"], t(fx, f"g8_{ci}_mask"))
        assert torch.equal(d["block_attention_mask"], t(fx, f"g8_{ci}_bmask"))
        assert d["initial_block_padding"] == int(fx[f"g8_{ci}_pad"])
        rv = o.revert_block_input_ids_to_vanilla(d["input_ids"], d["initial_block_padding"], 0)
        assert torch.equal(rv, t(fx, f"g8_{ci}_revert"))
        ci += 1
    assert ci == 3


def test_oracle_cached_generate_equals_nocache_loop(golden):
    """The oracle's cached generate() (torch.cat KV growth) == the reference's no-cache loop (already G6), and its
    incremental block-decoder step == the matching prefill row (G3 as an oracle property)."""
    dims, sd, fx = golden("micro")
    o = oracle_from_dims(dims, sd)
    ids = t(fx, "g1_ids")
    B, nb, L = ids.shape
    emb = o.embed(ids).view(B, nb, dims.bd.hidden)
    ones = torch.ones(B, nb, dtype=torch.long)
    full, _ = o.block_decoder(emb, ones)
    part, past = o.block_decoder(emb[:, :nb - 1], ones[:, :nb - 1])
    step, _ = o.block_decoder(emb[:, nb - 1:], ones, past)
    assert (step[:, 0] - full[:, -1]).abs().max() <= 1e-5


def test_eos_semantics_oracle(golden):
    """Q5/Q6: after EOS a row emits generation pad, later blocks are config pad, and the loop ends when all finish."""
    dims, sd, fx = golden("micro")
    base = oracle_from_dims(dims, sd)
    prompt = t(fx, "g6_0_prompt")
    ref = base.generate(prompt, max_length=23)
    eos = int(ref[0, 14])
    o = oracle_from_dims(dims, sd, gen_eos_token_id=eos, gen_pad_token_id=1)
    out = o.generate(prompt, max_length=23)
    assert out.shape == ref.shape
    row = out[0].tolist()
    k = row.index(eos, 10)
    assert row[:k + 1] == ref[0, :k + 1].tolist()
    assert all(v in (1, 0) for v in row[k + 1:])


def test_position_policies():
    assert td_position_table("train", 2, 4) == [0, 1, 2, 3, 4]
    assert td_position_table("hf439_generate", 2, 4) == [0, 0, 1, 2, 3]
    with pytest.raises(ValueError):
        td_position_table("nope", 2, 4)
