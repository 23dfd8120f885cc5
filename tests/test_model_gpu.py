"""Module-level and end-to-end parity (GPU): the MI355X modules, called through the reference's module API, against the
golden vectors captured from the reference's own classes and against the CPU oracle on the same seeded inputs.

fp32 instantiation: logits / hidden states within 1e-3 absolute of the reference CPU eager path (BASELINE.json
north_star tolerance; observed ~1e-5), greedy token ids bit-exact.  bf16 instantiation: looser tolerance + argmax
agreement (bf16 carries ~3 significant digits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import GOLDEN_CONFIGS, t  # noqa: E402
from block_transformer_amd.config import named_dims  # noqa: E402
from block_transformer_amd.model.utils import load_block_transformer_from_dims  # noqa: E402
from oracle.cpu_ref import oracle_from_dims  # noqa: E402
from tools.experiments.pipelined_engine import PipelinedEngine  # noqa: E402  (measured-negative experiment kept for its tests / A-B)
from oracle.cpu_ref import random_state_dict as _random_state_dict  # noqa: E402

_SD_CACHE = {}


def random_state_dict(dims, seed=0, **kw):
    """oracle.cpu_ref.random_state_dict, memoised for the session: ~40 tests draw the same 0.4 G-parameter 2-layer 1.2B-shaped
    weights (3-4 s of CPU randn each).  Tests only read the tensors (load_state_dict copies, casts make new tensors)."""
    key = (repr(dims), seed, tuple(sorted(kw.items())))
    if key not in _SD_CACHE:
        if len(_SD_CACHE) >= 10:
            _SD_CACHE.pop(next(iter(_SD_CACHE)))
        _SD_CACHE[key] = _random_state_dict(dims, seed=seed, **kw)
    return _SD_CACHE[key]


DEV = "cuda"
TOL = 1e-3          # north_star: logits within 1e-3 fp32
TIGHT = 2e-4        # what the fp32 kernels actually achieve on these sizes (guards against silent regressions)
# bf16 budget of the FULL-DEPTH (12 + 12 layer) comparisons against the fp32 oracle on the same bf16 weights: |error| <= BF16_REL *
# max|reference| + BF16_ABS.  Round 5 ratchet: four rounds of boxes observed 1.1-1.8 % of max (ctx 0.037-0.061,
# logits 0.044-0.057 at max|value| 3.4-4.2) against the old 3 % gate -- a kernel that doubled its error would still have passed.
BF16_REL, BF16_ABS = 2.2e-2, 1e-2


def maxabs(a, b):
    return (a.detach().float().cpu() - b.float()).abs().max().item()


def oracle_block_logits(o, prompt, block_tokens):
    """fp32 oracle logits of ONE generated block along the GPU's own tokens: context of `prompt` (+ earlier generated
    blocks already appended by the caller), token decoder teacher-forced on block_tokens[:, :-1] at the RoPE positions of
    cached generation.  Returns (ctx [B, H], logits [B, n, V]) with n = block_tokens.shape[1]."""
    d = o.preprocess_inputs_for_generation(prompt, None)
    ids, bm = d["input_ids"], d["block_attention_mask"]
    B, nb, L = ids.shape
    hidden, _ = o.block_decoder(o.embed(ids).view(B, nb, -1), bm)
    ctx = hidden[:, -1]
    n = block_tokens.shape[1]
    lg = o.token_decoder_teacher_forced(ctx, block_tokens[:, :n - 1], positions=o.td_pos)
    return ctx, lg


def assert_tokens_within_noise(ref_logits, tokens, rel, abs_, what=""):
    """Margin-aware greedy check (replaces "at least x % of the tokens agree"): every emitted token must be the oracle's
    argmax or lie inside the bf16 noise band below it -- oracle_logit[argmax] - oracle_logit[token] <= 2 * budget with
    budget = rel * max|logit| + abs_ (the logit error budget of the instantiation under test).  A wrong kernel emits
    tokens many budgets below the maximum; a bf16 near-tie stays inside the band.  Returns the fraction of exact matches."""
    budget = rel * ref_logits.abs().max().item() + abs_
    best = ref_logits.max(dim=-1).values
    chosen = ref_logits.gather(-1, tokens.unsqueeze(-1)).squeeze(-1)
    gap = (best - chosen)
    worst = gap.max().item()
    exact = (ref_logits.argmax(-1) == tokens).float().mean().item()
    print(f"[margin-aware {what}] worst gap {worst:.4f} vs band {2 * budget:.4f}; exact argmax agreement {exact:.3f}")
    assert worst <= 2 * budget, f"{what}: emitted token {worst:.4f} below the oracle's best logit (noise band {2 * budget:.4f})"
    return exact


@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
def test_embedder_g1(gpu_model, name):
    model, dims, sd, fx = gpu_model(name)
    ids = t(fx, "g1_ids").to(DEV)
    B, nb, L = ids.shape
    emb = model.embedder(ids.view(-1, L))
    assert emb.shape == (B * nb, 1, dims.bd.hidden)
    assert torch.equal(emb.view(B, nb, -1).cpu(), t(fx, "g1_emb"))
    eb = model.embed_blocks(ids, torch.ones_like(ids))
    assert eb.shape == (B, nb, 1, dims.bd.hidden)


@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
def test_block_decoder_prefill_g2(gpu_model, name):
    model, dims, sd, fx = gpu_model(name)
    x = t(fx, "g1_emb").to(DEV)
    B, nb, H = x.shape
    out = model.block_decoder(inputs_embeds=x, block_attention_mask=torch.ones(B, nb, dtype=torch.long, device=DEV),
                              return_dict=True)
    assert out.loss is None and out.logits is None and out.past_key_values.length == nb
    assert maxabs(out.hidden_states, t(fx, "g2_hidden")) <= TIGHT
    out2 = model.block_decoder(inputs_embeds=x, block_attention_mask=t(fx, "g2_mask").to(DEV), return_dict=True)
    d = (out2.hidden_states.cpu() - t(fx, "g2_hidden_masked")).abs()
    d[B - 1, 0] = 0     # fully masked query: unspecified in the reference
    assert d.max().item() <= TIGHT


@pytest.mark.parametrize("name", ["micro", "micro128", "micro64"])
def test_block_decoder_incremental_equals_prefill_g3(gpu_model, name):
    """Row t of a longer prefill == decode step t against the cache (global-KV append path)."""
    model, dims, sd, fx = gpu_model(name)
    x = t(fx, "g1_emb").to(DEV)
    B, nb, H = x.shape
    ones = torch.ones(B, nb, dtype=torch.long, device=DEV)
    full = model.block_decoder(inputs_embeds=x, block_attention_mask=ones).hidden_states
    part = model.block_decoder(inputs_embeds=x[:, :2].contiguous(), block_attention_mask=ones[:, :2])
    past = part.past_key_values
    assert maxabs(part.hidden_states, full[:, :2].cpu()) <= TIGHT
    for i in range(2, nb):
        step = model.block_decoder(inputs_embeds=x[:, i:i + 1].contiguous(), block_attention_mask=ones[:, :i + 1],
                                   past_key_values=past)
        past = step.past_key_values
        assert past.length == i + 1
        assert maxabs(step.hidden_states[:, 0], full[:, i].cpu()) <= TIGHT
        assert maxabs(step.hidden_states[:, 0], t(fx, "g2_hidden")[:, i]) <= TIGHT
    k0, v0 = past[0]
    assert k0.shape == (B, dims.bd.heads, nb, dims.bd.head_dim)


@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
def test_expansion_g4_and_token_decoder_g5(gpu_model, name):
    model, dims, sd, fx = gpu_model(name)
    ctx = t(fx, "g4_ctx").to(DEV)
    B = ctx.shape[0]
    exp = model.token_decoder.expand_block_embeddings(ctx[:, None, :])
    assert maxabs(exp, t(fx, "g4_expanded")) <= TIGHT
    toks = t(fx, "g5_tokens").to(DEV)
    tin = torch.cat([torch.zeros(B, 1, dtype=torch.long, device=DEV), toks], dim=1)
    out = model.token_decoder(input_ids=tin, block_embeddings=ctx[:, None, :], return_dict=True)
    lg = out.logits.cpu()
    assert lg.shape == (B, dims.L, dims.vocab)
    if "g5_logits" in fx:
        assert maxabs(lg, t(fx, "g5_logits")) <= TOL
        assert maxabs(lg, t(fx, "g5_logits")) <= TIGHT
    else:
        assert maxabs(lg[..., ::97], t(fx, "g5_logits_slice")) <= TOL
        assert torch.equal(lg.argmax(-1), t(fx, "g5_logits_argmax"))
    # G9: the same call with the position ids of the reference's cached generation routed through **kwargs
    from block_transformer_amd.engine import td_position_table
    pos9 = torch.tensor(td_position_table("hf439_generate", dims.E, dims.L))[None, :]
    lg9 = model.token_decoder(input_ids=tin, block_embeddings=ctx[:, None, :], return_dict=True,
                              position_ids=pos9).logits.cpu()
    if "g9_logits" in fx:
        assert maxabs(lg9, t(fx, "g9_logits")) <= TIGHT
        assert maxabs(lg9, t(fx, "g5_logits")) > 2e-4     # the two policies are distinguishable at this tolerance
    else:
        assert maxabs(lg9[..., ::97], t(fx, "g9_logits_slice")) <= TOL
        assert torch.equal(lg9.argmax(-1), t(fx, "g9_logits_argmax"))


def test_token_decoder_cached_steps_equal_teacher_forced(gpu_model):
    """forward(past=None, [BOS]) then cached single-token forwards == the teacher-forced logits (local-KV path)."""
    model, dims, sd, fx = gpu_model("micro")
    td = model.token_decoder
    ctx = t(fx, "g4_ctx").to(DEV)
    B = ctx.shape[0]
    toks = t(fx, "g5_tokens").to(DEV)
    ref = t(fx, "g5_logits")
    exp = td.expand_block_embeddings(ctx[:, None, :])
    bos = torch.zeros(B, 1, dtype=torch.long, device=DEV)
    out = td(input_ids=bos, expanded_block_embeddings=exp)
    assert maxabs(out.logits[:, 0], ref[:, 0]) <= TIGHT
    past = out.past_key_values
    for i in range(toks.shape[1]):
        am = torch.ones(B, 2 + i, dtype=torch.long, device=DEV)
        out = td(input_ids=toks[:, i:i + 1].contiguous(), attention_mask=am, expanded_block_embeddings=exp,
                 past_key_values=past)
        past = out.past_key_values
        assert maxabs(out.logits[:, 0], ref[:, i + 1]) <= TIGHT
    with pytest.raises(ValueError, match="BOS"):
        td(input_ids=toks[:, :1] + 1, block_embeddings=ctx[:, None, :])
    with pytest.raises(ValueError, match="xor"):
        td(input_ids=bos)


def test_token_decoder_generate(gpu_model):
    model, dims, sd, fx = gpu_model("micro")
    o = oracle_from_dims(dims, sd)
    ctx = t(fx, "g4_ctx")
    want = o.token_decoder_generate(ctx, dims.L)
    got = model.token_decoder.generate(block_embeddings=ctx[:, None, :].to(DEV), max_length=dims.L + 1)
    assert got.shape == (ctx.shape[0], dims.L + 1) and int(got[:, 0].abs().sum()) == 0
    assert torch.equal(got[:, 1:].cpu(), want)
    got2 = model.token_decoder.generate(block_embeddings=ctx[:, None, :].to(DEV), max_length=3)
    assert torch.equal(got2[:, 1:].cpu(), want[:, :2])


@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
@pytest.mark.parametrize("graphs", [True, False])
@pytest.mark.parametrize("policy,key", [("hf439_generate", "g9"), ("train", "g6")])
def test_generate_greedy_matches_reference_g6_g9(gpu_model, name, graphs, policy, key):
    """End-to-end greedy token ids == the reference's own classes (incl. P % L != 0 and D % L != 0): G9 = the default
    policy (token decoder at the position ids of the reference's cached generate()), G6 = plain arange positions."""
    model, dims, sd, fx = gpu_model(name)
    assert model.token_decoder.rope_policy == "hf439_generate"        # the default mirrors the reference's generate()
    model.use_graphs = graphs
    model.token_decoder.rope_policy = policy
    try:
        ci = 0
        while f"g6_{ci}_prompt" in fx:
            prompt = t(fx, f"g6_{ci}_prompt").to(DEV)
            ml = int(fx[f"g6_{ci}_max_length"])
            out = model.generate(prompt, max_length=ml)
            want = t(fx, f"{key}_{ci}_tokens")
            assert out.shape == want.shape, (out.shape, want.shape)
            assert torch.equal(out.cpu(), want), f"case {ci}"
            # context embeddings of the run (last block step) against the reference trace
            assert maxabs(model._engine.ctx, t(fx, f"{key}_{ci}_ctx")[-1]) <= TIGHT
            ci += 1
    finally:
        model.use_graphs = True
        model.token_decoder.rope_policy = "hf439_generate"


def test_generate_block_format_and_shapes(gpu_model):
    model, dims, sd, fx = gpu_model("micro")
    prompt = t(fx, "g6_1_prompt")            # P = 8 = 2 blocks
    B = prompt.shape[0]
    want = t(fx, "g9_1_tokens")
    ids = prompt.view(B, 2, 4).to(DEV)
    out = model.generate(ids, attention_mask=torch.ones_like(ids),
                         block_attention_mask=torch.ones(B, 2, dtype=torch.long, device=DEV), max_length=16)
    assert out.shape == (B, 4, 4) and torch.equal(out.view(B, -1).cpu(), want)
    with pytest.raises(ValueError, match="attention_mask must be specified"):
        model.generate(ids, block_attention_mask=torch.ones(B, 2, dtype=torch.long, device=DEV), max_length=16)


def test_generate_shard_equivalence(gpu_model):
    """Data-parallel correctness by construction: generating request shards separately == generating the batch."""
    model, dims, sd, fx = gpu_model("micro")
    prompt = torch.randint(1, dims.vocab, (6, 12), generator=torch.Generator().manual_seed(5)).to(DEV)
    whole = model.generate(prompt, max_length=28).cpu()
    parts = torch.cat([model.generate(prompt[i:i + 2].contiguous(), max_length=28).cpu() for i in (0, 2, 4)])
    assert torch.equal(whole, parts)
    o = oracle_from_dims(dims, sd)
    assert torch.equal(whole, o.generate(prompt.cpu(), max_length=28))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B", [130, 200, 256, 257])
def test_generate_batch_sizes_around_the_splitk_boundary(gpu_model, B, dtype):
    """The token decoder runs B*E prefix rows at step 0 and B rows afterwards, and rows <= 256 take the split-K GEMM path
    with its own scratch: batches with B <= 256 < B*E (and their neighbours) must size / use the workspace consistently
    (regression: B = 256 failed with 'workspace too small')."""
    model, dims, sd, fx = gpu_model("micro", dtype)
    prompt = torch.randint(1, dims.vocab, (B, 8), generator=torch.Generator().manual_seed(B)).to(DEV)
    out = model.generate(prompt, max_length=16).cpu()
    assert out.shape == (B, 16) and torch.equal(out[:, :8], prompt.cpu())
    # rows are independent: a slice of the batch generated on its own gives the same tokens
    part = model.generate(prompt[5:9].contiguous(), max_length=16).cpu()
    if dtype == torch.float32:
        assert torch.equal(out[5:9], part)
        o = oracle_from_dims(dims, sd)
        assert torch.equal(out[:3], o.generate(prompt[:3].cpu(), max_length=16))
    else:
        # bf16: the 4-row slice runs other kernels (GEMV / streamer) than the full batch -> a near-tie may flip one greedy
        # token and, from there on, that sequence's tail; most sequences must still agree completely
        assert (out[5:9] == part).all(dim=1).float().mean() >= 0.5 and (out[5:9, 8] == part[:, 8]).float().mean() >= 0.5


def test_generate_ragged_left_padded_batch(gpu_model):
    """Different prompt lengths, left-padded with an attention mask: fully masked leading blocks (Q4/Q5 masks)."""
    model, dims, sd, fx = gpu_model("micro")
    gen = torch.Generator().manual_seed(9)
    ids = torch.randint(1, dims.vocab, (3, 14), generator=gen)
    am = torch.ones_like(ids)
    am[1, :6] = 0       # 6 pad tokens: first block fully masked (after the 2-token left pad to 16)
    am[2, :3] = 0
    ids[am == 0] = 0
    o = oracle_from_dims(dims, sd)
    want = o.generate(ids, attention_mask=am, max_length=26)
    got = model.generate(ids.to(DEV), attention_mask=am.to(DEV), max_length=26)
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("cfg_pad,gen_pad", [(0, 1), (1, 1)])
def test_generate_eos_handling(gpu_model, cfg_pad, gen_pad):
    """Q5/Q6/Q7: EOS bookkeeping on device == the oracle's restatement of the reference loop."""
    model, dims, sd, fx = gpu_model("micro")
    prompt = t(fx, "g6_0_prompt")
    base = oracle_from_dims(dims, sd).generate(prompt, max_length=23)
    for col in (10, 13, 17):
        eos = int(base[0, col])
        o = oracle_from_dims(dims, sd, gen_eos_token_id=eos, gen_pad_token_id=gen_pad, pad_token_id=cfg_pad)
        want = o.generate(prompt, max_length=23)
        gcfg = model.token_decoder.generation_config
        old = (gcfg.eos_token_id, gcfg.pad_token_id, model.token_decoder.config.pad_token_id)
        try:
            gcfg.eos_token_id, gcfg.pad_token_id = eos, gen_pad
            model.token_decoder.config.pad_token_id = cfg_pad
            for graphs in (True, False):
                model.use_graphs = graphs
                got = model.generate(prompt.to(DEV), max_length=23)
                assert got.shape == want.shape, (got.shape, want.shape, col)
                assert torch.equal(got.cpu(), want), (col, graphs)
        finally:
            gcfg.eos_token_id, gcfg.pad_token_id, model.token_decoder.config.pad_token_id = old
            model.use_graphs = True


def test_generate_rope_policy_train_vs_oracle(gpu_model):
    """The non-default arange policy (config knob ``token_decoder.rope_policy=train``): engine == oracle."""
    model, dims, sd, fx = gpu_model("micro")
    prompt = t(fx, "g6_1_prompt")
    o = oracle_from_dims(dims, sd, rope_policy="train")
    want = o.generate(prompt, max_length=16)
    model.token_decoder.rope_policy = "train"
    try:
        got = model.generate(prompt.to(DEV), max_length=16)
    finally:
        model.token_decoder.rope_policy = "hf439_generate"
    assert torch.equal(got.cpu(), want)
    assert torch.equal(want, t(fx, "g6_1_tokens"))


def test_generate_streamer_protocol(gpu_model):
    model, dims, sd, fx = gpu_model("micro")

    class Rec:
        def __init__(self):
            self.events = []

        def put(self, v):
            self.events.append(("put", tuple(v.shape), v.clone()))

        def end(self):
            self.events.append(("end",))

    rec = Rec()
    prompt = t(fx, "g6_1_prompt").to(DEV)
    out = model.generate(prompt, max_length=16, streamer=rec)
    assert torch.equal(out.cpu(), t(fx, "g9_1_tokens"))
    kinds = [e[0] for e in rec.events]
    assert kinds == (["put"] * 5 + ["end"]) * 2           # per block: placeholder + L tokens, then end()
    B = prompt.shape[0]
    assert rec.events[0][1] == (B, 1) and rec.events[1][1] == (B,)
    streamed = torch.stack([e[2] for e in rec.events if e[0] == "put" and e[1] == (B,)], 1)
    assert torch.equal(streamed, out.cpu()[:, 8:])


def test_generate_sampling_runs(gpu_model):
    model, dims, sd, fx = gpu_model("micro")
    g = model.token_decoder.generation_config
    g.update(do_sample=True, temperature=0.8)
    try:
        prompt = t(fx, "g6_1_prompt").to(DEV)
        a = model.generate(prompt, max_length=24).cpu()
        assert a.shape == (prompt.shape[0], 24) and int(a.min()) >= 0 and int(a.max()) < dims.vocab
        assert torch.equal(a[:, :8], prompt.cpu())
    finally:
        g.update(do_sample=False, temperature=1.0)


@pytest.mark.parametrize("name", ["micro", "micro128", "micro64", "block_main_b4_5"])
def test_bf16_generate_tracks_fp32_oracle(gpu_model, name):
    """bf16 instantiation: first-block context embedding within bf16 tolerance of the fp32 oracle, and the greedy
    tokens of the first generated block agree wherever the oracle's top-1 margin exceeds the bf16 noise floor."""
    model, dims, sd, fx = gpu_model(name, torch.bfloat16)
    sd16 = {k: v.bfloat16().float() for k, v in sd.items()}      # oracle on the bf16-rounded weights
    o = oracle_from_dims(dims, sd16)
    prompt = t(fx, "g6_0_prompt")
    L = dims.L
    nb = -(-prompt.shape[1] // L)
    ml = nb * L + L
    want, trace = o.generate(prompt, max_length=ml, return_trace=True)
    got = model.generate(prompt.to(DEV), max_length=ml).cpu()
    assert got.shape == want.shape
    ctx = model._engine.ctx.float().cpu()
    ref = trace["ctx"][0]
    assert (ctx - ref).abs().max() <= 6e-2 * ref.abs().max() + 2e-2, (ctx - ref).abs().max()
    # margin-aware: along the GPU's own tokens every emitted token is the oracle's argmax or inside the bf16 noise band
    # (same relative budget as the context bound above), and the engine's last-step logits are inside that budget
    ref_ctx, ref_lg = oracle_block_logits(o, prompt, got[:, -L:])
    assert torch.allclose(ref_ctx, ref, atol=1e-5)
    assert_tokens_within_noise(ref_lg, got[:, -L:], 6e-2, 2e-2, f"{name} first block")
    e_lg = (model._engine.tok.logits.float().cpu() - ref_lg[:, -1]).abs().max().item()
    assert e_lg <= 6e-2 * ref_lg.abs().max().item() + 2e-2, e_lg


def test_real_dims_one_layer_bf16_logits():
    """1.2B-config tensor shapes (H=2048, d=128, rot=32, I=8192, V=50304) with ONE layer per decoder: exercises the
    MFMA GEMM at N in {6144, 2048, 8192, 4096, 50304}, the d=128 attention paths and the flash prefill, against the
    fp32 oracle on bf16-rounded weights."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b")
    one = dataclasses.replace(base, name="b4_1.2b_1layer", bd=dataclasses.replace(base.bd, layers=1),
                              td=dataclasses.replace(base.td, layers=1))
    sd = random_state_dict(one, seed=3, include_unused_embed_in=True)
    model = load_block_transformer_from_dims(one, "fp32")
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
    o = oracle_from_dims(one, sd16)
    prompt = torch.randint(0, one.vocab, (3, 40), generator=torch.Generator().manual_seed(1))
    want, trace = o.generate(prompt, max_length=48, return_trace=True)
    got = model.generate(prompt.to(DEV), max_length=48).cpu()
    assert got.shape == (3, 48) and torch.equal(got[:, :40], prompt)
    # first generated block: context + first-token logits
    eng = model._engine
    ref_ctx = trace["ctx"][-1]
    # the last context embedding depends on the first generated block's tokens: compare it for the sequences whose greedy
    # tokens of that block agree with the oracle (a bf16 near-tie may flip one on random-init weights)
    same = (got[:, 40:44] == want[:, 40:44]).all(dim=1)
    assert same.float().mean() >= 0.5, same
    d_ctx = (eng.ctx.float().cpu() - ref_ctx)[same].abs().max()
    assert d_ctx <= 6e-2 * ref_ctx.abs().max() + 2e-2, d_ctx
    _, lg = o.token_decoder_generate(trace["ctx"][0], 1, return_logits=True)
    out = model.token_decoder(input_ids=torch.zeros(3, 1, dtype=torch.long, device=DEV),
                              block_embeddings=trace["ctx"][0][:, None, :].bfloat16().to(DEV))
    d = (out.logits[:, 0].float().cpu() - lg[:, 0]).abs().max()
    assert d <= 5e-2 * lg.abs().max() + 2e-2, d
    model.release_buffers()


def test_fp32_logits_within_1e3_of_reference_real_head_dim(gpu_model):
    """north_star tolerance check at the production head size (d=128, rot=32): fp32 logits within 1e-3."""
    model, dims, sd, fx = gpu_model("micro128")
    ctx = t(fx, "g4_ctx").to(DEV)
    toks = t(fx, "g5_tokens").to(DEV)
    tin = torch.cat([torch.zeros(ctx.shape[0], 1, dtype=torch.long, device=DEV), toks], dim=1)
    lg = model.token_decoder(input_ids=tin, block_embeddings=ctx[:, None, :]).logits
    assert maxabs(lg, t(fx, "g5_logits")) <= TOL


@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
def test_forward_scoring_path_g7(gpu_model, name):
    """BlockTransformer.forward (teacher-forced logits [B, nb-1, L, V]) against the reference's own forward (G7):
    pins the block-shift alignment 'context of block i decodes block i+1'."""
    model, dims, sd, fx = gpu_model(name)
    ids = t(fx, "g1_ids").to(DEV)
    B, nb, L = ids.shape
    out = model(input_ids=ids, attention_mask=torch.ones_like(ids),
                block_attention_mask=torch.ones(B, nb, dtype=torch.long, device=DEV))
    lg = out["logits"].cpu()
    assert lg.shape == (B, nb - 1, L, dims.vocab) and out["loss"] is None
    if "g7_logits" in fx:
        assert maxabs(lg, t(fx, "g7_logits")) <= TOL
        assert maxabs(lg, t(fx, "g7_logits")) <= TIGHT
    else:
        assert maxabs(lg[..., ::97], t(fx, "g7_logits_slice")) <= TOL
        assert torch.equal(lg.argmax(-1), t(fx, "g7_logits_argmax"))


def test_forward_scoring_path_padding_semantics(gpu_model):
    """Right-padded batch (reference docstring, block_transformer.py:58-82): fully padded blocks are skipped and filled
    with eos_token_id, logits after the last content token of a block are zero, content logits match the oracle."""
    model, dims, sd, fx = gpu_model("micro")
    ids = t(fx, "g1_ids").clone()
    B, nb, L = ids.shape
    am = torch.ones_like(ids)
    am[1, 3, 2:] = 0      # block 3 of row 1 ends after 2 tokens (partial EOS block)
    am[1, 4, :] = 0       # block 4 of row 1 is pure padding
    ids[am == 0] = 0      # pad id == eos id == 0
    bm = am.any(-1).long()
    out = model(input_ids=ids.to(DEV), attention_mask=am.to(DEV), block_attention_mask=bm.to(DEV))["logits"].cpu()
    o = oracle_from_dims(dims, sd)
    ref = o.forward_logits(ids, am, bm)
    assert out.shape == ref.shape
    assert torch.all(out[1, 3] == model.token_decoder.config.eos_token_id)        # skipped padding block (index nb-2)
    # block 3 of row 1 (decoded from context 2): inputs [BOS, a, b | pad pad] -> logits for a^, b^ (+ the position that
    # predicts the token after b, which the reference keeps because attention_mask marks b as content) ; rest zero
    assert maxabs(out[1, 2, :2], ref[1, 2, :2]) <= TIGHT
    assert float(out[1, 2, 3].abs().max()) == 0.0
    assert maxabs(out[0], ref[0]) <= TIGHT and maxabs(out[1, :2], ref[1, :2]) <= TIGHT
    with pytest.raises(NotImplementedError):
        model(input_ids=ids.to(DEV), attention_mask=am.to(DEV), block_attention_mask=bm.to(DEV), labels=ids.to(DEV))


def test_checkpoint_roundtrip_safetensors(gpu_model, tmp_path):
    """results/<name>/checkpoint-N/model.safetensors with the reference key names loads into the MI355X modules (the
    newest checkpoint wins, the unused block_decoder embed_in may be absent) and reproduces the golden tokens."""
    from safetensors.torch import save_file
    from block_transformer_amd import inference_demo as demo
    model, dims, sd, fx = gpu_model("micro")
    root = tmp_path / "ckpts" / "micro"
    for step, scale in ((100, 0.5), (2500, 1.0)):
        d = root / f"checkpoint-{step}"
        d.mkdir(parents=True)
        save_file({k: (v * scale).contiguous() for k, v in sd.items() if k != "block_decoder.gpt_neox.embed_in.weight"},
                  str(d / "model.safetensors"))
    assert demo.get_checkpoint_path("micro", str(tmp_path / "ckpts")).endswith("checkpoint-2500/model.safetensors")
    from safetensors.torch import load_file
    fresh = load_block_transformer_from_dims(dims, "fp32").to(DEV)
    missing, unexpected = fresh.load_state_dict(load_file(demo.get_checkpoint_path("micro", str(tmp_path / "ckpts")),
                                                          device=DEV), strict=False)
    assert missing == ["block_decoder.gpt_neox.embed_in.weight"] and not unexpected
    fresh.token_decoder.generation_config.eos_token_id = None
    out = fresh.generate(t(fx, "g6_1_prompt").to(DEV), max_length=16)
    assert torch.equal(out.cpu(), t(fx, "g9_1_tokens"))


_SMOKE_MODELS = {}


@pytest.mark.parametrize("name", ["block_main_b4_5", "block_main_b4_19", "block_main_b4_85", "block_main_b4_300",
                                  "block_main_b4_800", "block_main_b4_1.2b", "block_main_b8_1.2b"])
@pytest.mark.parametrize("B,P,D", [(3, 38, 10), (8, 24, 8), (13, 16, 8), (100, 24, 8), (300, 16, 8)])
def test_generate_every_reference_config_smoke(name, B, P, D):
    """Every `block_main_*` architecture of the reference at odd batch sizes / prompt lengths (left pad, partial last
    block, M not a multiple of any tile: the persistent GEMM's edge tiles, the 128^2 / split-K kernels, and -- B = 3 / 8 / 13
    -- the fused GEMV and MFMA-streamer launch sequences of small stacks, in real use),
    bf16, random init: prompt preserved, ids in range, deterministic, hipGraph replay == eager launches, and rows
    independent of the batch they ride in."""
    import dataclasses
    dims = named_dims(name)
    # two layers per decoder keep the random-init 1.2B-class models cheap; widths / heads / head_dim are the real ones
    dims = dataclasses.replace(dims, bd=dataclasses.replace(dims.bd, layers=min(dims.bd.layers, 2)),
                               td=dataclasses.replace(dims.td, layers=min(dims.td.layers, 2)))
    model = _SMOKE_MODELS.get(name)
    if model is None:       # one bf16 model per architecture for the five (B, P, D) cases (building it is 4 of the 5 seconds)
        sd = random_state_dict(dims, seed=11, include_unused_embed_in=True)
        model = load_block_transformer_from_dims(dims, "fp32")
        model.load_state_dict(sd, strict=True)
        model = model.bfloat16().to(DEV)
        model.token_decoder.generation_config.eos_token_id = None
        _SMOKE_MODELS[name] = model
    model._engine = None
    prompt = torch.randint(1, dims.vocab, (B, P), generator=torch.Generator().manual_seed(B + P)).to(DEV)
    out = model.generate(prompt, max_length=P + D)
    pad = (-P) % dims.L      # the reference counts max_length on the left-padded sequence and strips the pad afterwards
    assert out.shape == (B, P + D - pad) and torch.equal(out[:, :P], prompt)
    assert int(out.min()) >= 0 and int(out.max()) < dims.vocab
    again = model.generate(prompt, max_length=P + D)
    assert torch.equal(out, again), "generation must be deterministic"
    model.use_graphs = False
    try:
        eager = model.generate(prompt, max_length=P + D)
    finally:
        model.use_graphs = True
    assert torch.equal(out, eager), "hipGraph replay must equal eager launches"
    # request independence (bf16: different GEMM tile paths for different M may round differently -> allow rare flips)
    part = model.generate(prompt[:2].contiguous(), max_length=P + D)
    assert (out[:2] == part).float().mean() >= 0.8


@pytest.mark.parametrize("name", ["block_main_b4_19", "block_main_b4_85", "block_main_b4_300", "block_main_b4_800"])
@pytest.mark.parametrize("B", [8, 300])
def test_every_reference_width_bf16_first_block_vs_oracle(name, B):
    """VERDICT r5 missing #4: the reference's other `block_main_b4_*` widths (conf/trainer/block_main_b4_*.yaml: H = 512 / 768 / 1024 /
    1536 ... with their own head counts, head_dim and intermediate sizes; util/config.py:92-105) against the ORACLE, not by property
    only -- the GEMM dispatch (GEMV, MFMA streamer, slab kernels, 128^2 family, persistent 256^2) has row-count regimes whose
    thresholds were tuned at H = 2048.  Two layers per decoder, bf16, P = 16: context embedding of the prompt and the logits of the
    whole first block along the GPU's own tokens vs the fp32 oracle on the bf16-rounded weights (2 rows; rows are independent), the
    budget of the shallow-model bf16 tests (6 % of max + 0.02), tokens margin-aware.  B = 8 runs the slab path's 8-row regime (block
    step) / 16 rows (token step 0), B = 300 the 257 .. 512-row regime (or the large-M sequence where a width fails its preconditions)
    and 600 rows at token step 0."""
    import dataclasses
    dims = named_dims(name)
    dims = dataclasses.replace(dims, bd=dataclasses.replace(dims.bd, layers=2), td=dataclasses.replace(dims.td, layers=2))
    key = ("oracle", name)
    if key not in _SMOKE_MODELS:
        sd = random_state_dict(dims, seed=17, include_unused_embed_in=True)
        model = load_block_transformer_from_dims(dims, "fp32")
        model.load_state_dict(sd, strict=True)
        model = model.bfloat16().to(DEV)
        model.token_decoder.generation_config.eos_token_id = None
        model.keep_logits = True
        _SMOKE_MODELS[key] = (model, oracle_from_dims(dims, {k: v.bfloat16().float() for k, v in sd.items()}))
    model, o = _SMOKE_MODELS[key]
    model._engine = None
    P, L = 16, dims.L
    prompt = torch.randint(1, dims.vocab, (B, P), generator=torch.Generator().manual_seed(7 * B + P))
    got = model.generate(prompt.to(DEV), max_length=P + L).cpu()
    eng = model._engine
    rows = [0, B - 1]
    ref_ctx, ref_lg = oracle_block_logits(o, prompt[rows], got[rows, P:P + L])
    ctx = eng.ctx.float().cpu()[rows]
    e_ctx = (ctx - ref_ctx).abs().max().item()
    e_lg = (eng.tok.logits.float().cpu()[rows] - ref_lg[:, -1]).abs().max().item()
    b_ctx, b_lg = 6e-2 * ref_ctx.abs().max().item() + 2e-2, 6e-2 * ref_lg.abs().max().item() + 2e-2
    print(f"[{name} B={B}] ctx err {e_ctx:.4f} / {b_ctx:.4f}, last-step logit err {e_lg:.4f} / {b_lg:.4f}")
    assert e_ctx <= b_ctx and e_lg <= b_lg
    assert_tokens_within_noise(ref_lg, got[rows, P:P + L], 6e-2, 2e-2, f"{name} B={B} first block")
    assert torch.equal(got[:, :P], prompt)
    model.release_buffers()


def test_full_size_1p2b_generate_properties():
    """BASELINE config at full depth (12 + 12 layers, P = 2048, bf16, random init): size-independent properties of the
    decode path -- prompt preserved, deterministic, hipGraph replay == eager launches, request shards generated
    separately agree with the batch (weights replicated, no cross-request state)."""
    dims = named_dims("block_main_b4_1.2b")
    sd = random_state_dict(dims, seed=21, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    B, P, D = 6, 2048, 16
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(2)).to(DEV)
    out = model.generate(prompt, max_length=P + D)
    assert out.shape == (B, P + D) and torch.equal(out[:, :P], prompt)
    assert torch.equal(out, model.generate(prompt, max_length=P + D))
    model.use_graphs = False
    try:
        assert torch.equal(out, model.generate(prompt, max_length=P + D))
    finally:
        model.use_graphs = True
    parts = torch.cat([model.generate(prompt[i:i + 2].contiguous(), max_length=P + D) for i in (0, 2, 4)])
    assert (parts == out).float().mean() >= 0.9     # bf16: M-dependent GEMM tiling may flip a near-tie
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B", [1, 3, 4])
def test_small_batch_fused_stack_matches_unfused(B, dbg_lib):
    """1..4 sequences: the decoder stacks run LayerNorm / RoPE / KV append inside the weight-streaming GEMVs (5 launches
    per layer).  Same weights, same prompt: the fused and the unfused (bt_debug_no_few_rows_fusion) launch sequences
    produce the same greedy tokens up to rare bf16 near-ties, and the same context embedding within bf16 rounding."""
    import dataclasses
    from block_transformer_amd._lib import lib
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=5, include_unused_embed_in=True)
    model = load_block_transformer_from_dims(dims, "fp32")
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(0, dims.vocab, (B, 64), generator=torch.Generator().manual_seed(B)).to(DEV)
    fused = model.generate(prompt, max_length=96).cpu()
    ctx_f = model._engine.ctx.float().cpu().clone()
    lib().bt_debug_no_few_rows_fusion(1)
    try:
        model._engine = None        # captured graphs hold the fused launch sequence
        unfused = model.generate(prompt, max_length=96).cpu()
        ctx_u = model._engine.ctx.float().cpu().clone()
    finally:
        lib().bt_debug_no_few_rows_fusion(0)
        model._engine = None
    assert (fused == unfused).float().mean() >= 0.9, (fused == unfused).float().mean()
    same = (fused == unfused).all(dim=1)
    if same.any():   # sequences whose tokens agree went through identical inputs: their last context must agree closely
        d = (ctx_f[same] - ctx_u[same]).abs().max() / ctx_u[same].abs().max()
        assert d <= 3e-2, d


@pytest.mark.parametrize("B,P", [(300, 32), (80, 512)])
def test_large_stack_single_residual_gemm_matches_two(B, P, dbg_lib):
    """> 64 rows: attention.dense and mlp.dense_4h_to_h as ONE GEMM over K = I + H (bt_linear2; bt_debug_lin2_large -- measured
    slower than the two launches, so the product keeps two: DESIGN.md section 3).  Same weights and prompt through both
    sequences: same greedy tokens up to rare bf16 near-ties, same last context embedding within bf16 rounding for the rows
    whose tokens agree.
    (80, 512): the prefill stack (80 x 128 rows) takes the persistent 256^2 kernel; (300, 32): the small-tile family."""
    import dataclasses
    from block_transformer_amd._lib import lib
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=5, include_unused_embed_in=True)
    model = load_block_transformer_from_dims(dims, "fp32")
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(B)).to(DEV)
    two = model.generate(prompt, max_length=P + 16).cpu()
    ctx_2 = model._engine.ctx.float().cpu().clone()
    lib().bt_debug_lin2_large(1)
    try:
        model._engine = None
        one = model.generate(prompt, max_length=P + 16).cpu()
        ctx_1 = model._engine.ctx.float().cpu().clone()
    finally:
        lib().bt_debug_lin2_large(0)
        model._engine = None
    assert (one == two).float().mean() >= 0.9, (one == two).float().mean()
    same = (one == two).all(dim=1)
    assert same.any()
    d = (ctx_1[same] - ctx_2[same]).abs().max() / ctx_2[same].abs().max()
    assert d <= 3e-2, d
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B", [6, 16])
def test_mid_batch_streamer_stack_matches_unfused(B, dbg_lib):
    """5..16 rows: the stacks run on the MFMA weight streamer with LayerNorm / RoPE / KV append / the second residual
    projection folded in.  Same weights and prompt through the unfused launch sequence (bt_debug_few_rows_max(4) +
    bt_debug_skinny_rows(4)): same greedy tokens up to rare bf16 near-ties."""
    import dataclasses
    from block_transformer_amd._lib import lib
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=5, include_unused_embed_in=True)
    model = load_block_transformer_from_dims(dims, "fp32")
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(0, dims.vocab, (B, 32), generator=torch.Generator().manual_seed(B)).to(DEV)
    fused = model.generate(prompt, max_length=48).cpu()
    lib().bt_debug_few_rows_max(4)
    lib().bt_debug_skinny_rows(4)
    try:
        model._engine = None
        unfused = model.generate(prompt, max_length=48).cpu()
    finally:
        lib().bt_debug_few_rows_max(16)
        lib().bt_debug_skinny_rows(16)
        model._engine = None
    assert torch.equal(fused[:, :32], prompt.cpu())
    # margin-aware instead of "x % of the rows agree": BOTH launch sequences must emit, at every step of the first block, the
    # fp32 oracle's argmax or a token inside the bf16 noise band below it (2-layer bf16 stack: 3 % of max|logit| + 0.01)
    sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
    o = oracle_from_dims(dims, sd16)
    for name, out in (("fused", fused), ("unfused", unfused)):
        _, ref_lg = oracle_block_logits(o, prompt.cpu(), out[:, 32:36])
        assert_tokens_within_noise(ref_lg, out[:, 32:36], 3e-2, 1e-2, f"streamer B={B} {name}")
    assert (fused[:, 32] == unfused[:, 32]).float().mean() >= 0.8


# ---- round 2: parity on the path that is benchmarked, long decodes, the half-batch pipeline, the harness entry points ------------
def _oracle_first_token(o, prompt):
    """(ctx [B, H], first-token logits [B, V]) of the fp32 oracle for a prompt batch."""
    d = o.preprocess_inputs_for_generation(prompt, None)
    ids, bm = d["input_ids"], d["block_attention_mask"]
    B, nb, L = ids.shape
    x = o.embed(ids).view(B, nb, -1)
    hidden, _ = o.block_decoder(x, bm)
    ctx = hidden[:, -1]
    _, lg = o.token_decoder_generate(ctx, 1, return_logits=True)
    return ctx, lg[:, 0]


_FULL = {}


def _full_1p2b():
    """block_main_b4_1.2b at full depth, bf16, random init directly on the GPU (HF-style N(0, 0.02) + non-zero biases), and
    the fp32 CPU oracle on exactly those bf16 weights.  Built once per test session (2.4 G parameters)."""
    if not _FULL:
        dims = named_dims("block_main_b4_1.2b")
        torch.manual_seed(21)
        with torch.device(DEV):
            model = load_block_transformer_from_dims(dims, "bf16", keep_unused_embed_in=False)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith(".bias"):
                    p.normal_(0.0, 0.02)
        model.token_decoder.generation_config.eos_token_id = None
        model.keep_logits = True
        sd16 = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        _FULL["m"] = (model, dims, oracle_from_dims(dims, sd16))
    return _FULL["m"]


@pytest.mark.parametrize("B", [2, 64])
def test_full_depth_1p2b_bf16_first_block_logits_vs_oracle(B):
    """BASELINE configs[1] at FULL depth (12 + 12 layers, H = 2048, V = 50304), bf16, P = 2048, through generate():
    B = 64 puts the prefill on the persistent 256x256 MFMA GEMM with the fused QKV/RoPE/append epilogue and the flash
    kernel (32768 rows per chunk), the decode on the 128^2 / split-K kernels; B = 2 on the skinny streamers.  Context
    embedding and first-token logits against the fp32 CPU oracle on the same (bf16) weights (oracle on 2 of the B
    sequences: rows are independent).
    bf16 budget: 24 layers of bf16-rounded activations (2^-9 relative per rounding, ~10 roundings per layer,
    accumulating as a random walk) -> |error| <= 2.2 % of max|value| + 0.01 absolute (BF16_REL / BF16_ABS; observed on MI355X: 1.1-1.8 % of
    max: ctx 0.041 / 0.061, logits 0.048 / 0.057 at max|value| 3.4-4.2); the greedy token must agree wherever the oracle's
    top-1 margin exceeds twice that budget."""
    model, dims, o = _full_1p2b()
    model.use_graphs = True
    P = 2048
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(B))
    got = model.generate(prompt.to(DEV), max_length=P + 1).cpu()       # one token: the engine keeps its logits
    eng = model._engine
    rows = [0, B - 1]
    ctx = eng.ctx.float().cpu()[rows]
    logits = eng.tok.logits.float().cpu()[rows]
    ref_ctx, ref_lg = _oracle_first_token(o, prompt[rows])
    budget_ctx = BF16_REL * ref_ctx.abs().max().item() + BF16_ABS
    budget_lg = BF16_REL * ref_lg.abs().max().item() + BF16_ABS
    e_ctx, e_lg = (ctx - ref_ctx).abs().max().item(), (logits - ref_lg).abs().max().item()
    print(f"[full-depth bf16 B={B}] ctx err {e_ctx:.4f} (budget {budget_ctx:.4f}, max {ref_ctx.abs().max():.3f}); "
          f"logit err {e_lg:.4f} (budget {budget_lg:.4f}, max {ref_lg.abs().max():.3f})")
    assert e_ctx <= budget_ctx and e_lg <= budget_lg
    top2 = ref_lg.topk(2, dim=-1).values
    margin = top2[:, 0] - top2[:, 1]
    for i, r in enumerate(rows):
        if margin[i] > 2 * budget_lg:
            assert int(got[r, P]) == int(ref_lg[i].argmax())
        assert int(got[r, P]) == int(logits[i].argmax())               # the emitted token is the argmax of the engine's logits
    assert torch.equal(got[:, :P], prompt)
    model.release_buffers()


def test_full_depth_1p2b_bf16_benchmark_batch_decode_dispatch_vs_oracle():
    """The benchmark's own dispatch: BASELINE configs[1] at FULL depth, bf16, **B = 4096** (the batch
    bench.py runs), short prompt (P = 64 -> 16 block positions, so the oracle and the KV cache stay small).  At this batch
    the decode runs the persistent 256^2 kernel (gemm_bf16_mfma256pp2_kernel) with the fused QKV / RoPE / append epilogue
    (M = 4096, N = 6144: 384 tiles, round-3 dispatch), the one-tile 256 x 128 kernel for the residual projections (N = 2048), the
    persistent kernel for h->4h / logits (+ fused argmax) and -- token step 0, q = 2 -> M = 8192 -- for everything; prefill =
    65536 rows through the persistent kernel + flash.  Three checkpoints, each against the fp32 CPU
    oracle on the same bf16 weights for 3 rows (rows are independent), same margin-aware budget as the B = 2 / 64 test
    (2.2 % of max + 0.01):
      (1) max_length = P + 1: context embedding + first-token logits (token step 0: q = 2, M = 8192);
      (2) max_length = P + 2: logits of token step 1 (q = 1, M = 4096: the 128^2 fused-QKV path), oracle teacher-forced on
          the GPU's own first token;
      (3) max_length = P + L + 1: context + logits of the SECOND block (block-decoder decode step: q = 1, M = 4096 against
          the global KV written by the prefill + roll-over embedding), oracle prefilled on prompt + the GPU's own first block."""
    model, dims, o = _full_1p2b()
    model.use_graphs = True
    B, P, L = 4096, 64, dims.L
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(4096))
    rows = [0, 1777, B - 1]
    dev_prompt = prompt.to(DEV)

    def budget(ref):
        return BF16_REL * ref.abs().max().item() + BF16_ABS

    # (1) first block, token step 0
    got = model.generate(dev_prompt, max_length=P + 1).cpu()
    eng = model._engine
    ctx, lg = eng.ctx.float().cpu()[rows], eng.tok.logits.float().cpu()[rows]
    ref_ctx, ref_lg = _oracle_first_token(o, prompt[rows])
    e_ctx, e_lg = (ctx - ref_ctx).abs().max().item(), (lg - ref_lg).abs().max().item()
    print(f"[B=4096 step 0] ctx err {e_ctx:.4f} / {budget(ref_ctx):.4f}, logit err {e_lg:.4f} / {budget(ref_lg):.4f}")
    assert e_ctx <= budget(ref_ctx) and e_lg <= budget(ref_lg)
    assert torch.equal(got[rows, P], lg.argmax(-1))
    assert_tokens_within_noise(ref_lg[:, None], got[rows, P:P + 1], BF16_REL, BF16_ABS, "B=4096 step 0")
    assert torch.equal(got[:, :P], prompt)
    # (2) token step 1 (q = 1): teacher-forced oracle along the GPU's first token
    got2 = model.generate(dev_prompt, max_length=P + 2).cpu()
    assert torch.equal(got2[:, :P + 1], got), "first token changed between two runs"
    lg1 = model._engine.tok.logits.float().cpu()[rows]
    _, ref_blk = oracle_block_logits(o, prompt[rows], got2[rows, P:P + 2])
    e1 = (lg1 - ref_blk[:, 1]).abs().max().item()
    print(f"[B=4096 step 1] logit err {e1:.4f} / {budget(ref_blk[:, 1]):.4f}")
    assert e1 <= budget(ref_blk[:, 1])
    assert torch.equal(got2[rows, P + 1], lg1.argmax(-1))
    assert_tokens_within_noise(ref_blk, got2[rows, P:P + 2], BF16_REL, BF16_ABS, "B=4096 steps 0-1")
    # (3) second block: block-decoder decode step over the global KV
    got3 = model.generate(dev_prompt, max_length=P + L + 1).cpu()
    assert torch.equal(got3[:, :P + 2], got2)
    eng = model._engine
    ctx2, lg2 = eng.ctx.float().cpu()[rows], eng.tok.logits.float().cpu()[rows]
    ref_ctx2, ref_lg2 = _oracle_first_token(o, got3[rows, :P + L])
    e_ctx2, e_lg2 = (ctx2 - ref_ctx2).abs().max().item(), (lg2 - ref_lg2).abs().max().item()
    print(f"[B=4096 block 2] ctx err {e_ctx2:.4f} / {budget(ref_ctx2):.4f}, logit err {e_lg2:.4f} / {budget(ref_lg2):.4f}")
    assert e_ctx2 <= budget(ref_ctx2) and e_lg2 <= budget(ref_lg2)
    assert torch.equal(got3[rows, P + L], lg2.argmax(-1))
    _, ref_blk1 = oracle_block_logits(o, prompt[rows], got3[rows, P:P + L])
    assert_tokens_within_noise(ref_blk1, got3[rows, P:P + L], BF16_REL, BF16_ABS, "B=4096 first block")
    # every row was decoded (not just the sampled three): ids in range, rows differ
    assert int(got3.min()) >= 0 and int(got3.max()) < dims.vocab
    assert len({tuple(r.tolist()) for r in got3[:64, P:]}) > 8
    model.release_buffers()


@pytest.mark.parametrize("B", [64, 512])
def test_full_depth_1p2b_bf16_steady_state_decode_vs_oracle(B):
    """the steady-state decode of the HEADLINE configuration against the oracle at model level, in bf16.  Full-depth
    block_main_b4_1.2b, P = 2048 (512 cached block positions from the flash prefill), max_length = P + L + 1: one complete token
    block, then bt_roll_over_enqueue -- embedder lookup of the GPU's own first block, RoPE + append at cache row 512 and the
    UNSPLIT q = 1 global attention over 513 cached positions in every layer (B * heads >= 512) -- then token step 0 of block 2.
    B = 64 takes the mid-batch split-K stack, B = 512 the large-M stack the benchmark runs (128^2 / 256x128 / persistent GEMMs with
    the fused QKV + RoPE + append epilogue).  Context of block 2 and its first logits for 2 rows against the fp32 oracle prefilled on
    prompt + the GPU's first block (same budget as the B = 4096 checkpoints: 2.2 % of max + 0.01), tokens margin-aware."""
    model, dims, o = _full_1p2b()
    model.use_graphs = True
    P, L = 2048, dims.L
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(1000 + B))
    got = model.generate(prompt.to(DEV), max_length=P + L + 1).cpu()
    eng = model._engine
    assert eng.n_split == 1 and int(eng.n_cur.item()) == P // L + 1
    rows = [0, B - 1]
    ctx2, lg2 = eng.ctx.float().cpu()[rows], eng.tok.logits.float().cpu()[rows]
    ref_ctx2, ref_lg2 = _oracle_first_token(o, got[rows, :P + L])

    def budget(ref):
        return BF16_REL * ref.abs().max().item() + BF16_ABS

    e_ctx, e_lg = (ctx2 - ref_ctx2).abs().max().item(), (lg2 - ref_lg2).abs().max().item()
    print(f"[steady state B={B}: block 2 over 513 cached positions] ctx err {e_ctx:.4f} / {budget(ref_ctx2):.4f}, "
          f"logit err {e_lg:.4f} / {budget(ref_lg2):.4f}")
    assert e_ctx <= budget(ref_ctx2) and e_lg <= budget(ref_lg2)
    assert torch.equal(got[rows, P + L], lg2.argmax(-1))
    assert_tokens_within_noise(ref_lg2[:, None], got[rows, P + L:P + L + 1], BF16_REL, BF16_ABS, f"steady state B={B} block 2")
    _, ref_blk1 = oracle_block_logits(o, prompt[rows], got[rows, P:P + L])
    assert_tokens_within_noise(ref_blk1, got[rows, P:P + L], BF16_REL, BF16_ABS, f"steady state B={B} first block")
    assert torch.equal(got[:, :P], prompt)
    assert int(got.min()) >= 0 and int(got.max()) < dims.vocab
    model.release_buffers()


def test_full_depth_1p2b_bf16_p2048_d2048_last_block_vs_oracle():
    """VERDICT r5 missing #3 / next #3a: BASELINE configs[3]'s geometry against the oracle where it was checked by property only.
    Full-depth block_main_b4_1.2b, bf16, B = 64, P = 2048, D = 2048: 512 prompt positions through the flash prefill, then 511 block
    steps (roll-over: embedder lookup of the GPU's own block, RoPE + append, UNSPLIT q = 1 global attention -- B * heads = 1024 >= 512
    -- over 513 ... 1023 cached rows of the n_max = 1024 cache, model/block_transformer.py:370-426 at D = 2048) on hipGraph replays,
    stopping on the FIRST token of the last block (max_length = P + D - L + 1).  The context embedding of that last block and its
    first-token logits for 2 rows against the fp32 oracle prefilled on prompt + every token the GPU generated before it (1023 block
    positions), same budget as every full-depth bf16 check (2.2 % of max + 0.01); the emitted token margin-aware."""
    model, dims, o = _full_1p2b()
    model.use_graphs = True
    B, P, D, L = 64, 2048, 2048, dims.L
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(4096))
    ml = P + D - L + 1
    got = model.generate(prompt.to(DEV), max_length=ml).cpu()
    eng = model._engine
    assert got.shape == (B, ml) and eng.n_split == 1 and int(eng.n_cur.item()) == (P + D) // L - 1
    rows = [0, B - 1]
    ctx, lg = eng.ctx.float().cpu()[rows], eng.tok.logits.float().cpu()[rows]
    ref_ctx, ref_lg = _oracle_first_token(o, got[rows, :ml - 1])

    def budget(ref):
        return BF16_REL * ref.abs().max().item() + BF16_ABS

    e_ctx, e_lg = (ctx - ref_ctx).abs().max().item(), (lg - ref_lg).abs().max().item()
    print(f"[P2048/D2048 B={B}: last block over {(P + D) // L - 1} cached positions] ctx err {e_ctx:.4f} / {budget(ref_ctx):.4f}, "
          f"logit err {e_lg:.4f} / {budget(ref_lg):.4f}")
    assert e_ctx <= budget(ref_ctx) and e_lg <= budget(ref_lg)
    assert torch.equal(got[rows, ml - 1], lg.argmax(-1))
    assert_tokens_within_noise(ref_lg[:, None], got[rows, ml - 1:ml], BF16_REL, BF16_ABS, "P2048/D2048 last block")
    assert torch.equal(got[:, :P], prompt) and int(got.min()) >= 0 and int(got.max()) < dims.vocab
    assert len({tuple(r.tolist()) for r in got[:, P:P + 64]}) > 8          # rows decode different things
    model.release_buffers()


def test_full_depth_1p2b_fp32_logits_within_1e3_of_oracle():
    """The north-star tolerance, literally, on the north-star model (VERDICT r4 missing #1): "logits within 1e-3 fp32 of the reference
    CPU eager path" on the fp32 instantiation of FULL-DEPTH block_main_b4_1.2b (12 + 12 layers, H = 2048, I = 8192: K = 8192
    accumulations in every dense_4h_to_h, V = 50304; 9.7 GB of fp32 weights), B = 2, P = 2048 (512 block positions through the
    row-wise prefill attention), max_length = P + L + 1: context embedding and logits (model/token_decoder/base.py:349) of block 1 and
    of block 2 (roll-over + global attention over 513 cached positions) within 1e-3 of the oracle, every token equal."""
    dims = named_dims("block_main_b4_1.2b")
    torch.manual_seed(23)
    with torch.device(DEV):
        model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".bias"):
                p.normal_(0.0, 0.02)
    model.token_decoder.generation_config.eos_token_id = None
    model.keep_logits = True
    model.use_graphs = True
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    o = oracle_from_dims(dims, sd)
    B, P, L = 2, 2048, dims.L
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(2048))
    try:
        got1 = model.generate(prompt.to(DEV), max_length=P + 1).cpu()
        eng = model._engine
        ctx1, lg1 = eng.ctx.float().cpu(), eng.tok.logits.float().cpu()
        ref_ctx1, ref_lg1 = _oracle_first_token(o, prompt)
        e_ctx1, e_lg1 = (ctx1 - ref_ctx1).abs().max().item(), (lg1 - ref_lg1).abs().max().item()
        print(f"[full-depth fp32 block 1] ctx err {e_ctx1:.2e} (max {ref_ctx1.abs().max():.3f}), logit err {e_lg1:.2e} "
              f"(max {ref_lg1.abs().max():.3f})")
        assert e_ctx1 <= TOL and e_lg1 <= TOL, (e_ctx1, e_lg1)
        assert torch.equal(got1[:, P], ref_lg1.argmax(-1))
        got = model.generate(prompt.to(DEV), max_length=P + L + 1).cpu()
        eng = model._engine
        assert int(eng.n_cur.item()) == P // L + 1
        ctx2, lg2 = eng.ctx.float().cpu(), eng.tok.logits.float().cpu()
        ref_ctx2, ref_lg2 = _oracle_first_token(o, got[:, :P + L])
        e_ctx2, e_lg2 = (ctx2 - ref_ctx2).abs().max().item(), (lg2 - ref_lg2).abs().max().item()
        print(f"[full-depth fp32 block 2 over 513 cached positions] ctx err {e_ctx2:.2e}, logit err {e_lg2:.2e}")
        assert e_ctx2 <= TOL and e_lg2 <= TOL, (e_ctx2, e_lg2)
        # every logit of the first block along the GPU's own tokens (token steps 1 .. L-1 over the local window) + the tokens themselves
        _, ref_blk = oracle_block_logits(o, prompt, got[:, P:P + L])
        assert torch.equal(got[:, P:P + L], ref_blk.argmax(-1))
        assert torch.equal(got[:, P + L], ref_lg2.argmax(-1))
        assert torch.equal(got[:, :P + 1], got1)
    finally:
        model.release_buffers()
        del model
        torch.cuda.empty_cache()


_FULL_B8 = {}


def _full_b8():
    """block_main_b8_1.2b (BASELINE configs[4]: L = 8, embedder width 256, local window of 9 keys) at full depth, bf16, random init on
    the GPU, and the fp32 CPU oracle on exactly those bf16 weights.  Built once per session."""
    if not _FULL_B8:
        dims = named_dims("block_main_b8_1.2b")
        torch.manual_seed(29)
        with torch.device(DEV):
            model = load_block_transformer_from_dims(dims, "bf16", keep_unused_embed_in=False)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith(".bias"):
                    p.normal_(0.0, 0.02)
        model.token_decoder.generation_config.eos_token_id = None
        model.keep_logits = True
        sd16 = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        _FULL_B8["m"] = (model, dims, oracle_from_dims(dims, sd16))
    return _FULL_B8["m"]


@pytest.mark.parametrize("B", [64, 512])
def test_full_depth_b8_1p2b_bf16_first_block_and_steady_state_vs_oracle(B):
    """BASELINE configs[4] at its OWN size (VERDICT r4 missing #2): full-depth block_main_b8_1.2b, bf16, P = 2048 -> 256 prompt
    positions of L = 8 tokens (embedder rows of width h_e = 256 concatenated: model/embedder/base.py:55-61), token-decoder window of
    E + L - 1 = 9 keys, 8 token steps per block.  B = 64 takes the mid-batch split-K stack (folded local attention, KMAX = 12), B = 512
    the large-M stack of the benchmark.  Two checkpoints against the fp32 oracle on the same bf16 weights (2 rows), budget
    BF16_REL * max + BF16_ABS:
      (1) max_length = P + 1: context embedding + first-token logits;
      (2) max_length = P + L + 1: the whole first block margin-aware along the GPU's own tokens (token steps 1 .. 7 read 3 .. 9 local
          keys), then roll-over (lookup of 8 ids x 256 columns, append at cache row 256, unsplit attention over 257 positions) and
          context + first logits of block 2."""
    model, dims, o = _full_b8()
    assert dims.L == 8
    model.use_graphs = True
    P, L = 2048, dims.L
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(8000 + B))
    rows = [0, B - 1]

    def budget(ref):
        return BF16_REL * ref.abs().max().item() + BF16_ABS

    got1 = model.generate(prompt.to(DEV), max_length=P + 1).cpu()
    eng = model._engine
    ctx, lg = eng.ctx.float().cpu()[rows], eng.tok.logits.float().cpu()[rows]
    ref_ctx, ref_lg = _oracle_first_token(o, prompt[rows])
    e_ctx, e_lg = (ctx - ref_ctx).abs().max().item(), (lg - ref_lg).abs().max().item()
    print(f"[b8 full depth B={B} block 1] ctx err {e_ctx:.4f} / {budget(ref_ctx):.4f}, logit err {e_lg:.4f} / {budget(ref_lg):.4f}")
    assert e_ctx <= budget(ref_ctx) and e_lg <= budget(ref_lg)
    assert torch.equal(got1[rows, P], lg.argmax(-1))
    got = model.generate(prompt.to(DEV), max_length=P + L + 1).cpu()
    eng = model._engine
    assert int(eng.n_cur.item()) == P // L + 1
    assert torch.equal(got[:, :P + 1], got1)
    ctx2, lg2 = eng.ctx.float().cpu()[rows], eng.tok.logits.float().cpu()[rows]
    ref_ctx2, ref_lg2 = _oracle_first_token(o, got[rows, :P + L])
    e_ctx2, e_lg2 = (ctx2 - ref_ctx2).abs().max().item(), (lg2 - ref_lg2).abs().max().item()
    print(f"[b8 full depth B={B} block 2 over 257 cached positions] ctx err {e_ctx2:.4f} / {budget(ref_ctx2):.4f}, "
          f"logit err {e_lg2:.4f} / {budget(ref_lg2):.4f}")
    assert e_ctx2 <= budget(ref_ctx2) and e_lg2 <= budget(ref_lg2)
    assert torch.equal(got[rows, P + L], lg2.argmax(-1))
    assert_tokens_within_noise(ref_lg2[:, None], got[rows, P + L:P + L + 1], BF16_REL, BF16_ABS, f"b8 B={B} block 2")
    _, ref_blk1 = oracle_block_logits(o, prompt[rows], got[rows, P:P + L])
    assert_tokens_within_noise(ref_blk1, got[rows, P:P + L], BF16_REL, BF16_ABS, f"b8 B={B} first block (8 token steps)")
    assert torch.equal(got[:, :P], prompt)
    assert int(got.min()) >= 0 and int(got.max()) < dims.vocab
    model.release_buffers()


def test_full_size_1p2b_p2048_d2048_per_gpu_properties():
    """BASELINE configs[3] per GPU (the 8x line shards requests, one shard per GPU: what ONE rank runs): block_main_b4_1.2b
    bf16, B = 8, P = 2048, D = 2048 -> n_max = 1024 block positions: the flash prefill v3 <-> v2 switch (prompts <= 1024
    positions take v3; here 512) with a cache sized for 1024, 512 block-step replays with the global attention growing
    512 -> 1023 keys and the split-N factor re-chosen for n_max = 1024.  Size-independent properties: prompt preserved, ids in
    range, deterministic replay, graphs on == graphs off bit for bit."""
    model, dims, _ = _full_1p2b()
    model.use_graphs = True
    B, P, D = 8, 2048, 2048
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(33)).to(DEV)
    a = model.generate(prompt, max_length=P + D).cpu()
    assert a.shape == (B, P + D) and torch.equal(a[:, :P], prompt.cpu())
    assert int(a.min()) >= 0 and int(a.max()) < dims.vocab
    eng = model._engine
    assert eng.n_max >= 1024 and eng.n_split > 1, (eng.n_max, eng.n_split)
    b = model.generate(prompt, max_length=P + D).cpu()
    assert torch.equal(a, b), "graph replay is not deterministic / state leaked between runs"
    model.use_graphs = False
    try:
        c = model.generate(prompt, max_length=P + D).cpu()
    finally:
        model.use_graphs = True
    assert torch.equal(a, c), "captured graphs differ from the eager launch sequence"
    assert len({tuple(r.tolist()) for r in a[:, P:P + 64]}) > 1
    model.release_buffers()


def test_long_prompt_long_decode_fp32_tokens_vs_oracle(gpu_model):
    """configs[3]-class shape in fp32 against the oracle: micro128 (d = 128, rot = 32), P = 2048 (512 prompt positions through
    the flash prefill), D = 1024 (256 block steps: cache grows 512 -> 767 of n_max = 768, split-N attention under graph
    replay): greedy tokens bit-exact."""
    model, dims, sd, fx = gpu_model("micro128")
    o = oracle_from_dims(dims, sd)
    B, P, D = 2, 2048, 1024
    prompt = torch.randint(1, dims.vocab, (B, P), generator=torch.Generator().manual_seed(77))
    want = o.generate(prompt, max_length=P + D)
    model.use_graphs = True
    got = model.generate(prompt.to(DEV), max_length=P + D).cpu()
    assert model._engine.n_max >= (P + D) // dims.L and model._engine.n_split > 1
    first_bad = (got != want).any(0).nonzero()
    assert torch.equal(got, want), f"first differing column {int(first_bad[0]) if len(first_bad) else None}"


def test_greedy_tokens_do_not_depend_on_keeping_logits():
    """BlockTransformer.keep_logits = False (the default: the logits GEMM's epilogue feeds the argmax reduction, no [B, V] matrix is
    written) == keep_logits = True, bit for bit, graphs on and off, on the 2-layer 1.2B-shaped bf16 model at B = 512 (fused
    epilogue) and B = 5 (fallback: GEMV + argmax); with keep_logits = False the engine's logits buffer is never touched."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=6, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    assert model.keep_logits is False
    for B in (512, 5):
        prompt = torch.randint(0, dims.vocab, (B, 32), generator=torch.Generator().manual_seed(B)).to(DEV)
        outs = {}
        for keep in (True, False):
            for graphs in (True, False):
                model.keep_logits, model.use_graphs = keep, graphs
                model._engine = None
                out = model.generate(prompt, max_length=48).cpu()
                outs[(keep, graphs)] = out
                if not keep and B == 512:
                    eng = model._engine
                    with torch.inference_mode():
                        eng.tok.logits.fill_(float("nan"))
                    again = model.generate(prompt, max_length=48).cpu()
                    assert torch.equal(again, out)
                    assert bool(torch.isnan(eng.tok.logits.float()).all()), "logits were written although keep_logits is False"
        first = outs[(True, True)]
        assert all(torch.equal(v, first) for v in outs.values())
    model.keep_logits, model.use_graphs = False, True
    del model
    torch.cuda.empty_cache()


def _two_group_eos_case(model, dims, B, P, D, seed):
    """A batch whose even rows share one prompt and whose odd rows share another, and an EOS id that the even rows emit as the
    first token of their THIRD generated block and nobody emits before: with that EOS half of the batch finishes exactly there."""
    g = torch.Generator().manual_seed(seed)
    for _ in range(50):
        pa, pb = torch.randint(1, dims.vocab, (2, P), generator=g)
        prompt = torch.stack([pa if i % 2 == 0 else pb for i in range(B)])
        model.token_decoder.generation_config.eos_token_id = None
        free = model.generate(prompt[:2].to(DEV), max_length=P + D).cpu()
        L = dims.L
        eos = int(free[0, P + 2 * L])
        if eos not in free[0, P:P + 2 * L].tolist() and eos not in free[1, P:].tolist():
            return prompt, eos, free
    raise AssertionError("no usable EOS id found")


@pytest.mark.parametrize("name,dtype", [("micro", torch.float32), ("micro128", torch.float32), ("micro", torch.bfloat16)])
def test_compaction_of_finished_sequences_matches_masked_engine(gpu_model, name, dtype):
    """Reference model/block_transformer.py:381-398,421 (only decode_mask rows are decoded): with EOS handling on, once half of the
    batch has finished the engine continues on a compacted batch.  Tokens == the masked engine (compact_finished = False) ==
    the CPU oracle (fp32), graphs on and off; and the compaction really happened (192 -> 96 rows after the third block)."""
    model, dims, sd, fx = gpu_model(name, dtype)
    B, P, D = 192, 16, 48
    prompt, eos, free = _two_group_eos_case(model, dims, B, P, D, seed=3)
    gcfg = model.token_decoder.generation_config
    try:
        gcfg.eos_token_id = eos
        outs = {}
        for compact in (True, False):
            for graphs in (True, False):
                model.compact_finished, model.use_graphs = compact, graphs
                out = model.generate(prompt.to(DEV), max_length=P + D).cpu()
                outs[(compact, graphs)] = out
                if compact:
                    comp = model._engine.compactions
                    assert comp and comp[0][1] == B // 2 and comp[0][2] == B // 2, comp
                else:
                    assert model._engine.compactions == []
        first = outs[(True, True)]
        assert all(torch.equal(v, first) for v in outs.values())
        # even rows: free-running tokens up to and including the EOS, pad afterwards; odd rows: untouched by their neighbours
        L = dims.L
        pad = int(model.token_decoder.config.pad_token_id)
        assert torch.equal(first[0, :P + 2 * L + 1], free[0, :P + 2 * L + 1])
        assert set(first[0, P + 3 * L:].tolist()) <= {pad}
        assert torch.equal(first[1], free[1, :first.shape[1]]) or eos in free[1].tolist()
        if dtype == torch.float32:
            o = oracle_from_dims(dims, sd, gen_eos_token_id=eos, gen_pad_token_id=gcfg.pad_token_id,
                                 pad_token_id=int(model.token_decoder.config.pad_token_id))
            want = o.generate(prompt[:4], max_length=P + D)
            assert torch.equal(first[:4, :want.shape[1]], want)
    finally:
        gcfg.eos_token_id = None
        model.compact_finished, model.use_graphs = True, True


def _staged_eos_case(model, dims, P, D, n_cand=96, seed=11):
    """Three prompts and an EOS id such that, greedy: prompt A emits it first inside its 2nd..3rd generated block, prompt B
    first at least two blocks later, prompt C never within D tokens.  Found by one free-running generate over random prompts."""
    L = dims.L
    g = torch.Generator().manual_seed(seed)
    cand = torch.randint(1, dims.vocab, (n_cand, P), generator=g)
    model.token_decoder.generation_config.eos_token_id = None
    free = model.generate(cand.to(DEV), max_length=P + D).cpu()
    gen = free[:, P:]
    nblk = gen.shape[1] // L
    for v in range(1, dims.vocab):
        hit = (gen[:, :nblk * L] == v).view(n_cand, nblk, L).any(-1)          # [cand, block]
        first = torch.where(hit.any(1), hit.float().argmax(1), torch.full((n_cand,), 10 ** 6))
        a = ((first >= 1) & (first <= 2)).nonzero().flatten()
        never = (first >= 10 ** 6).nonzero().flatten()
        if not len(a) or not len(never):
            continue
        fa = int(first[a[0]])
        b = ((first >= fa + 2) & (first < nblk - 2)).nonzero().flatten()
        if len(b):
            ia, ib, ic = int(a[0]), int(b[0]), int(never[0])
            return (cand[ia], cand[ib], cand[ic]), v, (free[ia], free[ib], free[ic]), (fa, int(first[ib]))
    raise AssertionError("no staged EOS case found")


@pytest.mark.parametrize("graphs", [True, False])
def test_compaction_padded_and_nested_matches_masked_engine(gpu_model, graphs):
    """(1) a live count that is NOT a multiple of the 32-row bucket -- 70 of 192 rows continue on a 96-row view with
    26 inert padding rows (finished, row_map -1); (2) a second compaction of that view (30 of its rows continue on a 32-row
    view: row_map composes).  Tokens == the masked engine == the CPU oracle; a second generate() on the same (cached) views
    reproduces them."""
    model, dims, sd, fx = gpu_model("micro", torch.float32)
    P, D, L = 16, 64, dims.L
    (pa, pb, pc), eos, frees, (blk_a, blk_b) = _staged_eos_case(model, dims, P, D)
    nA, nB, nC = 122, 40, 30
    order = torch.randperm(nA + nB + nC, generator=torch.Generator().manual_seed(1))
    prompt = torch.stack([pa] * nA + [pb] * nB + [pc] * nC)[order]
    group = torch.tensor([0] * nA + [1] * nB + [2] * nC)[order]
    gcfg = model.token_decoder.generation_config
    try:
        gcfg.eos_token_id = eos
        model.use_graphs = graphs
        model.compact_finished = True
        out = model.generate(prompt.to(DEV), max_length=P + D).cpu()
        comp = model._engine.compactions
        assert [(c[1], c[2]) for c in comp] == [(nB + nC, 96), (nC, 32)], comp
        again = model.generate(prompt.to(DEV), max_length=P + D).cpu()
        assert torch.equal(again, out), "cached compacted views leaked state between generate() calls"
        model.compact_finished = False
        masked = model.generate(prompt.to(DEV), max_length=P + D).cpu()
        assert model._engine.compactions == []
        assert torch.equal(out, masked)
        pad = int(model.token_decoder.config.pad_token_id)
        for gi, free in enumerate(frees):   # a row equals its free-running tokens up to its EOS; group C never finishes
            row = out[int((group == gi).nonzero()[0, 0])]
            hits = (free[P:] == eos).nonzero().flatten()
            upto = P + int(hits[0]) + 1 if len(hits) else row.shape[0]
            assert torch.equal(row[:upto], free[:upto]), gi
            if gi < 2:
                assert len(hits) and set(row[P + (int(hits[0]) // L + 1) * L:].tolist()) <= {pad}, gi
        o = oracle_from_dims(dims, sd, gen_eos_token_id=eos, gen_pad_token_id=gcfg.pad_token_id, pad_token_id=pad)
        pick = [int((group == gi).nonzero()[0, 0]) for gi in range(3)]
        want = o.generate(prompt[pick], max_length=P + D)
        assert torch.equal(out[pick][:, :want.shape[1]], want)
        assert int(out.min()) >= 0 and int(out.max()) < dims.vocab
    finally:
        gcfg.eos_token_id = None
        model.compact_finished, model.use_graphs = True, True


@pytest.mark.parametrize("stream", [False, True])
def test_compaction_under_sampling_and_streamer(gpu_model, stream):
    """The mode the reference's own caller runs (inference_demo.py:62-83: do_sample + streamer, EOS on): the sampler keys its RNG
    by the ORIGINAL row (bt_engine_desc.row_map), so for a fixed seed the compacted continuation draws exactly what the masked
    engine draws.  With a streamer the compacted engine streams the live rows only (the reference's token decoder sees
    decode_mask rows); the emitted ids are the same."""
    model, dims, sd, fx = gpu_model("micro", torch.float32)
    B, P, D, L = 128, 16, 64, dims.L
    prompt = torch.randint(1, dims.vocab, (B, P), generator=torch.Generator().manual_seed(9))
    gcfg = model.token_decoder.generation_config
    gcfg.update(do_sample=True, temperature=1.3)
    gcfg.seed = 1234

    class Rec:
        def __init__(self):
            self.shapes, self.blocks = [], []          # blocks: (original rows, [token tensors]) per block

        def put(self, v):
            self.shapes.append(tuple(v.shape))
            if v.dim() == 2:                           # the per-block placeholder: bt_rows was published just before it
                self.blocks.append((self.bt_rows.clone(), []))
            else:
                self.blocks[-1][1].append(v.clone())

        def end(self):
            pass

    try:
        gcfg.eos_token_id = None
        free = model.generate(prompt.to(DEV), max_length=P + D).cpu()[:, P:]
        # an EOS id that about half of the rows draw within the first half of the decode
        half = free[:, :D // 2]
        frac = torch.stack([(half == v).any(1).float().mean() for v in range(dims.vocab)])
        eos = int((frac - 0.6).abs().argmin())
        assert 0.3 < float(frac[eos]) < 0.9, float(frac[eos])
        gcfg.eos_token_id = eos
        outs, recs = {}, {}
        for compact in (True, False):
            model.compact_finished = compact
            rec = Rec() if stream else None
            outs[compact] = model.generate(prompt.to(DEV), max_length=P + D, streamer=rec).cpu()
            recs[compact] = rec
            if compact:
                comp = list(model._engine.compactions)
                assert comp, "no compaction happened"
        assert torch.equal(outs[True], outs[False])
        if stream:
            # ONE protocol: with or without compaction the streamer gets the reference's decode_mask rows of every block
            # (model/block_transformer.py:389-398) -- the sequences unfinished when the block started, ascending original order --
            # and their original row indices; both engines stream the same rows and the same tokens
            widths = [sh[0] for sh in recs[True].shapes]
            assert widths[0] == B and min(widths) < B // 2 + 1, "finished sequences are not streamed"
            assert recs[True].shapes == recs[False].shapes
            full = outs[True]
            unfinished = torch.ones(B, dtype=torch.bool)
            for bi, ((rows_c, toks_c), (rows_m, toks_m)) in enumerate(zip(recs[True].blocks, recs[False].blocks)):
                assert torch.equal(rows_c, rows_m) and torch.equal(rows_c, unfinished.nonzero().flatten()), f"block {bi}: rows"
                blk = full[:, P + bi * L:P + (bi + 1) * L]
                for ti, (a, b) in enumerate(zip(toks_c, toks_m)):
                    assert torch.equal(a, b), f"block {bi} token {ti}: compacted and masked engines streamed different tokens"
                # the streamed values are the raw draws; the output holds them up to and including a row's EOS (pad after it)
                got = torch.stack(toks_c, 1)
                want = blk[rows_c]
                seen_eos = torch.zeros(len(rows_c), dtype=torch.bool)
                for ti in range(got.shape[1]):
                    keep = ~seen_eos
                    assert torch.equal(got[keep, ti], want[keep, ti]), f"block {bi} token {ti}: stream != output for attributed rows"
                    seen_eos |= got[:, ti] == eos
                unfinished &= ~(blk == eos).any(1)
    finally:
        gcfg.update(do_sample=False, temperature=1.0)
        gcfg.eos_token_id = None
        gcfg.seed = 0
        model.compact_finished = True


def test_device_guard_flags_a_replay_past_the_cache(gpu_model):
    """a C binder that replays its captured block step once too often must not write past the global KV cache.
    With bt_engine_desc.err_flag set, the roll-over clamps *n_cur and raises the sticky CACHE_FULL bit; bt_finish_block refuses a
    block index past out_ids (BLOCK_RANGE).  The Python engine surfaces the bits as BtError."""
    import ctypes
    from block_transformer_amd import _lib
    model, dims, sd, fx = gpu_model("micro", torch.float32)
    prompt = t(fx, "g6_1_prompt").to(DEV)
    out = model.generate(prompt, max_length=16)
    eng = model._engine
    guard_k = eng.glob.k.clone()
    assert int(eng.err_flag.item()) == 0
    with torch.inference_mode(), torch.cuda.stream(eng.stream):
        eng.n_cur.fill_(eng.n_max)                                  # "one replay too many"
        _lib.check(_lib.lib().bt_roll_over_enqueue(ctypes.byref(eng.desc), _lib.current_stream()), "roll")
    torch.cuda.synchronize()
    assert int(eng.err_flag.item()) & _lib.BT_DEV_ERR_CACHE_FULL
    assert int(eng.n_cur.item()) == eng.n_max                       # clamped to n_max - 1 for the step, bumped back
    assert torch.equal(eng.glob.k[:, :, :, :eng.n_max - 1], guard_k[:, :, :, :eng.n_max - 1])   # only the last row was touched
    with pytest.raises(_lib.BtError, match="cache full"):
        eng.check_device_errors()
    assert int(eng.err_flag.item()) == 0
    with torch.inference_mode(), torch.cuda.stream(eng.stream):
        eng.n_cur.fill_(eng.NB)                                     # block index past out_ids
        before = eng.out_ids.clone()
        _lib.check(_lib.lib().bt_token_block_enqueue(ctypes.byref(eng.desc), 0, dims.L, dims.L, _lib.current_stream()), "tok")
    torch.cuda.synchronize()
    assert int(eng.err_flag.item()) & _lib.BT_DEV_ERR_BLOCK_RANGE and torch.equal(eng.out_ids, before)
    with torch.inference_mode():
        eng.err_flag.zero_()
    again = model.generate(prompt, max_length=16)
    assert torch.equal(again, out)


def test_stream_k_handoff_timeout_makes_generate_raise(dbg_lib):
    """the stream-K GEMM's bounded hand-off must fail LOUDLY.  Two-layer block_main_b4_1.2b (H = 2048), B = 4096: every
    decoder step runs the fused residual projections as one stream-K GEMM whose 128 tiles are each shared by two workgroups.  With the
    debug hook that makes workgroup 8 skip its publish (consumers give up after 2000 polls) generate() raises BtError -- the tokens
    are not returned -- through bt_attn_args.err_flag -> bt_engine_desc.err_flag (BT_DEV_ERR_HANDOFF_TIMEOUT); the scoring path
    (decoder module forward -> DecoderRuntime.check_errors) raises too; and without the hook the same engine generates again."""
    import dataclasses
    from block_transformer_amd import _lib
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_1layer", bd=dataclasses.replace(base.bd, layers=1),
                               td=dataclasses.replace(base.td, layers=1))
    torch.manual_seed(3)
    with torch.device(DEV):
        model = load_block_transformer_from_dims(dims, "bf16", keep_unused_embed_in=False)
    model.token_decoder.generation_config.eos_token_id = None
    B, P = 4096, 8
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(1)).to(DEV)
    model.use_graphs = False                 # (a captured graph carries the kernel arguments of its capture, the hook included)
    good = model.generate(prompt, max_length=P + 8).cpu()
    emb = torch.randn(B, 1, dims.bd.hidden, device=DEV, dtype=torch.bfloat16)       # 4096 rows through the block decoder MODULE
    _lib.lib().bt_debug_stream_k_drop(8, 2000)               # (even virtual ids publish at M = 4096: 2 workgroups per tile)
    try:
        with pytest.raises(_lib.BtError, match="hand-off timed out"):
            model.generate(prompt, max_length=P + 8)
        assert int(model._engine.err_flag.item()) == 0             # read and cleared by the raise
        with pytest.raises(_lib.BtError, match="hand-off timed out"):
            model.block_decoder(inputs_embeds=emb)
    finally:
        _lib.lib().bt_debug_stream_k_drop(-1, 0)
    assert torch.equal(model.generate(prompt, max_length=P + 8).cpu(), good)
    model.block_decoder(inputs_embeds=emb)
    del model
    torch.cuda.empty_cache()


@pytest.mark.perf
def test_compaction_speeds_up_an_eos_heavy_batch(dbg_lib):
    """2-layer 1.2B-shaped bf16 model, B = 1024, half of the rows finish at the start of the third
    generated block: the compacted continuation (512 rows) must be faster than masking the dead rows, with identical tokens.
    Wall-clock comparison inside the regular suite: the assertion is only the sanity bound t_c < t_m; the ratio is
    printed (and recorded by tools/gpu_runs) rather than gated.
    Round 5: a 512-row step now runs on the slab path (other K split than the 1024-row step's persistent GEMM), so in bf16 the
    continuation rounds differently from the masked engine: bit-identity is asserted with the slab path limited to 256 rows (both
    engines on the same row-independent kernels: what compaction itself must preserve), and under the default dispatch the runs must
    agree up to the compaction point, on every finished row, and on the first token after it except near-ties."""
    import dataclasses
    import time
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=8, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    B, P, D = 1024, 64, 128
    prompt, eos, _ = _two_group_eos_case(model, dims, B, P, D, seed=5)
    gcfg = model.token_decoder.generation_config
    gcfg.eos_token_id = eos
    times, outs = {}, {}
    try:
        for compact in (True, False, True, False):
            model.compact_finished = compact
            model.generate(prompt.to(DEV), max_length=P + D)            # warm-up: engines, graphs
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model.generate(prompt.to(DEV), max_length=P + D)
            torch.cuda.synchronize()
            times.setdefault(compact, []).append(time.perf_counter() - t0)
            outs[compact] = out.cpu()
        L = dims.L
        cut = P + 3 * L                                                  # the even rows emit EOS in the third block; the engine compacts behind it
        assert torch.equal(outs[True][:, :cut], outs[False][:, :cut])
        fin = outs[False][:, P + 2 * L] == eos
        assert int(fin.sum()) == B // 2
        assert torch.equal(outs[True][fin], outs[False][fin])
        # the live rows' first block after the compaction: every token the fp32 oracle's argmax or inside the bf16 noise band below it,
        # given the run's own prefix -- for the 512-row continuation (slab path) and for the masked 1024-row engine alike
        sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
        o = oracle_from_dims(dims, sd16)
        rows = torch.tensor([1, B - 1])
        for name, out in (("compacted", outs[True]), ("masked", outs[False])):
            _, ref_lg = oracle_block_logits(o, out[rows, :cut], out[rows, cut:cut + L])
            assert_tokens_within_noise(ref_lg, out[rows, cut:cut + L], 3e-2, 1e-2, f"compaction B={B} {name}")
        t_c, t_m = min(times[True]), min(times[False])
        print(f"[compaction] B={B}: compacted {t_c * 1e3:.1f} ms vs masked {t_m * 1e3:.1f} ms -> {t_m / t_c:.2f}x")
        assert t_c < t_m, (t_c, t_m)
        same = {}
        dbg_lib.bt_debug_mid_rows_max(256)                               # 512- and 1024-row steps on the same kernels
        for compact in (True, False):
            model.compact_finished = compact
            model._engine = None
            same[compact] = model.generate(prompt.to(DEV), max_length=P + D).cpu()
        assert torch.equal(same[True], same[False])
    finally:
        dbg_lib.bt_debug_mid_rows_max(512)
        gcfg.eos_token_id = None
        model.compact_finished = True
        model._engine = None
    del model
    torch.cuda.empty_cache()


def test_fp32_production_dims_one_layer_logits_within_1e3():
    """north_star tolerance at PRODUCTION tensor dims: fp32 instantiation, one layer per decoder, H = 2048, d = 128,
    rot = 32, I = 8192, V = 50304: context, first-token logits within 1e-3 of the CPU oracle, greedy tokens equal."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b")
    one = dataclasses.replace(base, name="b4_1.2b_1layer_fp32", bd=dataclasses.replace(base.bd, layers=1),
                              td=dataclasses.replace(base.td, layers=1))
    sd = random_state_dict(one, seed=4, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(one, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    o = oracle_from_dims(one, sd)
    prompt = torch.randint(0, one.vocab, (2, 64), generator=torch.Generator().manual_seed(2))
    got1 = model.generate(prompt.to(DEV), max_length=65).cpu()
    ref_ctx, ref_lg = _oracle_first_token(o, prompt)
    eng = model._engine
    assert maxabs(eng.ctx, ref_ctx) <= TOL, maxabs(eng.ctx, ref_ctx)
    assert maxabs(eng.tok.logits, ref_lg) <= TOL, maxabs(eng.tok.logits, ref_lg)
    assert torch.equal(got1[:, 64], ref_lg.argmax(-1))
    want = o.generate(prompt, max_length=76)
    got = model.generate(prompt.to(DEV), max_length=76).cpu()
    assert torch.equal(got, want)
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name,B,P,D", [("micro128", 2, 128, 1024), ("micro", 3, 126, 1030)])
def test_long_decode_graph_replay_matches_oracle(gpu_model, name, B, P, D):
    """BASELINE configs[2] shape (short prompt, LONG decode) in fp32 with graphs on: hundreds of block-step replays, n_cur
    growing from P/L past 256 cached positions so the split-N global attention (B*heads < 512, n_max >= 256) runs under
    replay; greedy tokens bit-exact against the CPU oracle, and graphs on == graphs off."""
    model, dims, sd, fx = gpu_model(name)
    o = oracle_from_dims(dims, sd)
    prompt = torch.randint(1, dims.vocab, (B, P), generator=torch.Generator().manual_seed(P))
    ml = -(-P // dims.L) * dims.L + D
    want = o.generate(prompt, max_length=ml)
    model.use_graphs = True
    got = model.generate(prompt.to(DEV), max_length=ml).cpu()
    eng = model._engine
    assert eng.n_split > 1 and eng.n_max >= 256, (eng.n_split, eng.n_max)
    assert got.shape == want.shape
    first_bad = (got != want).any(0).nonzero()
    assert torch.equal(got, want), f"first differing column {int(first_bad[0]) if len(first_bad) else None}"
    model.use_graphs = False
    try:
        eager = model.generate(prompt.to(DEV), max_length=ml).cpu()
    finally:
        model.use_graphs = True
    assert torch.equal(eager, got)


def test_full_size_1p2b_long_decode_properties():
    """BASELINE configs[2] at full size: block_main_b4_1.2b bf16, B = 8, P = 128, D = 2048 (512 block-step replays, the
    5..16-row MFMA streamer stacks, split-N global attention from 256 cached positions on).  Size-independent properties:
    prompt preserved, ids in range, replay is deterministic (second run identical: all per-run state is reset), and the
    captured graphs launch exactly the eager sequence (graphs on == graphs off, bit for bit)."""
    model, dims, _ = _full_1p2b()
    model.use_graphs = True
    B, P, D = 8, 128, 2048
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(11)).to(DEV)
    a = model.generate(prompt, max_length=P + D).cpu()
    assert a.shape == (B, P + D) and torch.equal(a[:, :P], prompt.cpu())
    assert int(a.min()) >= 0 and int(a.max()) < dims.vocab
    assert model._engine.n_split > 1
    b = model.generate(prompt, max_length=P + D).cpu()
    assert torch.equal(a, b), "graph replay is not deterministic / state leaked between runs"
    model.use_graphs = False
    c = model.generate(prompt, max_length=P + D).cpu()
    assert torch.equal(a, c), "captured graphs differ from the eager launch sequence"
    # sequences are independent: the 8 rows do not all produce the same text
    assert len({tuple(r.tolist()) for r in a[:, P:P + 64]}) > 1
    model.use_graphs = True
    model.release_buffers()


@pytest.mark.parametrize("B", [2, 5, 64])
@pytest.mark.parametrize("sync", ["0", "1"])
def test_half_batch_pipeline_equals_single_engine(gpu_model, B, sync, monkeypatch):
    """tools/experiments PipelinedEngine (two half-batch shards on two streams, one token block apart) == the single-stream engine,
    bit for bit in fp32 (sequences are independent), incl. odd batches, EOS handling and P % L != 0."""
    monkeypatch.setenv("BT_PIPELINE_SYNC", sync)
    model, dims, sd, fx = gpu_model("micro")
    prompt = torch.randint(1, dims.vocab, (B, 10), generator=torch.Generator().manual_seed(B)).to(DEV)
    try:
        model.engine_factory = None
        model.release_buffers()
        one = model.generate(prompt, max_length=39).cpu()
        assert type(model._engine).__name__ == "DecodeEngine"
        model.engine_factory = PipelinedEngine
        two = model.generate(prompt, max_length=39).cpu()
        assert model._engine.pipelined and len(model._engine.shards) == 2
        assert torch.equal(one, two)
        assert torch.equal(two, model.generate(prompt, max_length=39).cpu())      # replay of the captured shard graphs
        # EOS: pick a token the greedy run produces and make it EOS
        gcfg = model.token_decoder.generation_config
        eos = int(one[0, 14])
        old = (gcfg.eos_token_id, gcfg.pad_token_id)
        try:
            # with EOS handling on, HF's "stop the block once EVERY row has produced EOS" (Q6) is a whole-batch condition:
            # generate() falls back to the single-stream engine, so the result cannot depend on the sharding
            gcfg.eos_token_id, gcfg.pad_token_id = eos, 1
            two_e = model.generate(prompt, max_length=39).cpu()
            assert type(model._engine).__name__ == "DecodeEngine"
            model.engine_factory = None
            one_e = model.generate(prompt, max_length=39).cpu()
            assert torch.equal(one_e, two_e)
            o = oracle_from_dims(dims, sd, gen_eos_token_id=eos, gen_pad_token_id=1)
            assert torch.equal(one_e, o.generate(prompt.cpu(), max_length=39))
        finally:
            gcfg.eos_token_id, gcfg.pad_token_id = old
    finally:
        model.engine_factory = None
        model.release_buffers()


def test_half_batch_pipeline_bf16_full_width_layers():
    """The pipeline at production tensor widths (2-layer 1.2B-shaped model, bf16, B = 512 -> two 256-row shards): same
    greedy tokens as the single-stream engine up to bf16 near-ties (the shards run other GEMM tile shapes than the whole
    batch), first token of every sequence identical to the argmax of its own logits."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=5, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(0, dims.vocab, (512, 64), generator=torch.Generator().manual_seed(1)).to(DEV)
    model.engine_factory = None
    one = model.generate(prompt, max_length=96).cpu()
    model.engine_factory = PipelinedEngine
    two = model.generate(prompt, max_length=96).cpu()
    assert model._engine.pipelined
    assert torch.equal(two[:, :64], prompt.cpu())
    assert (one[:, 64] == two[:, 64]).float().mean() >= 0.97, (one[:, 64] == two[:, 64]).float().mean()
    # whole-row agreement is only a sanity bound: a near-tie flips one token and, from there on, that row's tail (observed 0.51)
    assert (one == two).all(dim=1).float().mean() >= 0.3, (one == two).all(dim=1).float().mean()
    del model
    torch.cuda.empty_cache()


def test_graphs_follow_reloaded_weights_and_seed(gpu_model):
    """the captured graphs bake in parameter pointers and the sampler seed.  Re-materialised parameters
    (state_dict round trip with assign=True, .cpu().cuda()) and a changed seed must not be replayed through stale graphs."""
    model, dims, sd, fx = gpu_model("micro")
    prompt = t(fx, "g6_1_prompt").to(DEV)
    base = model.generate(prompt, max_length=24).cpu()
    try:
        # new storage, different values: the output must follow the new weights
        sd2 = {k: (v.clone() * (1.5 if k.endswith("embed_out.weight") else 1.0)).to(DEV) for k, v in sd.items()}
        sd2["token_decoder.embed_out.weight"] = sd2["token_decoder.embed_out.weight"].flip(0).contiguous()
        model.load_state_dict(sd2, strict=True, assign=True)
        moved = model.generate(prompt, max_length=24).cpu()
        o2 = oracle_from_dims(dims, {k: v.cpu() for k, v in sd2.items()})
        assert torch.equal(moved, o2.generate(prompt.cpu(), max_length=24))
        assert not torch.equal(moved, base)
    finally:
        model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=True, assign=True)
    assert torch.equal(model.generate(prompt, max_length=24).cpu(), base)
    g = model.token_decoder.generation_config
    g.update(do_sample=True, temperature=1.3)
    try:
        outs = []
        for seed in (1, 2, 1):
            g.seed = seed
            outs.append(model.generate(prompt, max_length=40).cpu())
        assert torch.equal(outs[0], outs[2]), "same seed, same draws"
        assert not torch.equal(outs[0], outs[1]), "a new seed must reach the captured sampler"
        model.use_graphs = False
        g.seed = 1
        assert torch.equal(model.generate(prompt, max_length=40).cpu(), outs[0]), "graphs on == graphs off under sampling"
    finally:
        model.use_graphs = True
        g.update(do_sample=False, temperature=1.0)
        g.seed = 0


def test_topk_sampler_is_deterministic_and_bounded():
    """candidates are compacted in vocabulary order (no atomic slot race): the same (seed, step, row) draws the
    same token every time, also with many ties at the threshold; top_k beyond the 1024-candidate buffer is refused."""
    from block_transformer_amd import _lib, ops
    V = 50304
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(64, V, generator=g).bfloat16()              # bf16: plenty of exact ties
    logits[:8] = 0.0                                                 # every score tied: 50304 candidates at the threshold
    logits = logits.to(DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    outs = [ops.topk_sample(logits, temperature=0.9, top_k=50, seed=7, step=step).cpu() for _ in range(5)]
    for o_ in outs[1:]:
        assert torch.equal(o_, outs[0])
    assert int(outs[0].min()) >= 0 and int(outs[0].max()) < V
    assert int(outs[0][:8].max()) < 1024, "all-tied rows keep the lowest-index ties"
    # every draw lies in the top-k set (ties included) of its row
    f = logits.float().cpu()
    kth = f.topk(50, dim=-1).values[:, -1]
    picked = f[torch.arange(64), outs[0]]
    assert bool((picked >= kth).all())
    with pytest.raises(_lib.BtError, match="top_k"):
        ops.topk_sample(logits, temperature=1.0, top_k=5000, seed=1, step=step)


def test_generate_benchmark_printout(gpu_model, capsys):
    """generate(benchmark=True) prints the reference's per-iteration report (block_transformer.py:428-444)."""
    model, dims, sd, fx = gpu_model("micro")
    prompt = t(fx, "g6_1_prompt").to(DEV)
    out = model.generate(prompt, max_length=24, benchmark=True).cpu()
    text = capsys.readouterr().out
    assert "Block decoder times:" in text and "Token decoder times:" in text and "Total time:" in text
    tm = model.last_timers
    assert len(tm.iter_ms("token_decoder")) == 4 and len(tm.iter_ms("block_decoder")) == 3   # 16 new tokens = 4 blocks
    assert torch.equal(out, model.generate(prompt, max_length=24).cpu())


def test_measure_generation_time_main_runs(tmp_path):
    """The harness counterpart of measure_generation_time.py end to end on the GPU at micro size: fixed batch list, then
    the reference's auto-tune (B = 1, 2 -> linear memory model -> bisection) under an explicit memory bound."""
    from block_transformer_amd import measure_generation_time as mgt
    common = ["++block_decoder.config.num_hidden_layers=1", "++token_decoder.config.num_hidden_layers=1",
              "++benchmark_prefill_length=32", "++benchmark_decode_length=16", f"++save_dir={tmp_path}"]
    df = mgt.main(["--config-name", "block_main_b4_5", "++benchmark_batch_sizes=[2,4]", "++output_dir=fixed"] + common)
    assert list(df["batch_size"]) == [2, 4] and not df["oom"].any()
    assert (df["tokens_per_s"] > 0).all() and (tmp_path / "fixed" / "generation_time.csv").exists()
    # auto-tune under a bound a little above what B = 4 needed (weights + engine), stop rule 0.05 GiB free
    limit = float(df["max_memory_allocated"].max()) + 0.25
    df = mgt.main(["--config-name", "block_main_b4_5", "++output_dir=auto", f"++benchmark_memory_limit_gib={limit}",
                   "++benchmark_min_free_gib=0.05"] + common)
    bs = sorted(df["batch_size"])
    assert bs[:2] == [1, 2] and len(bs) >= 3 and bs[-1] > 2, bs
    assert float(df["max_memory_allocated"].max()) <= limit + 0.2


def test_measure_generation_time_profiling_writes_a_trace(tmp_path, capsys):
    """++profiling=true (reference measure_generation_time.py:381-390,150-155): the timed generate runs with the library's
    kernel timer, prints the family table and writes one Chrome-trace file per batch size; launches are ordered and
    disjoint in time (one stream), and their families are the ones a decode uses."""
    import json
    from block_transformer_amd import measure_generation_time as mgt
    df = mgt.main(["--config-name", "block_main_b4_5", "++benchmark_batch_sizes=[3]", "++output_dir=prof", "++profiling=true",
                   f"++log_path={tmp_path / 'traces'}", "++block_decoder.config.num_hidden_layers=1",
                   "++token_decoder.config.num_hidden_layers=1", "++benchmark_prefill_length=32",
                   "++benchmark_decode_length=16", f"++save_dir={tmp_path}"])
    assert list(df["batch_size"]) == [3]
    text = capsys.readouterr().out
    assert "kernel family" in text and "attn_rows" in text
    files = list((tmp_path / "traces").iterdir())
    assert len(files) == 1 and files[0].name.endswith("_3_32_16.log")
    ev = [e for e in json.load(open(files[0]))["traceEvents"] if e["ph"] == "X"]
    assert len(ev) > 50
    assert {"linear_mfma", "linear_generic"} & {e["name"] for e in ev} and "attn_rows" in {e["name"] for e in ev}
    for a, b in zip(ev, ev[1:]):
        assert b["ts"] >= a["ts"] and a["dur"] > 0
        assert b["ts"] + 1.0 >= a["ts"] + a["dur"], (a, b)     # 1 us slack: hipEvent timestamps are quantised


def test_inference_demo_main_runs(capsys):
    """The harness counterpart of inference_demo.py end to end (random init: no checkpoint / tokenizer on the box):
    sampling with temperature, streamer attached, throughput report."""
    from block_transformer_amd import inference_demo as demo
    out = demo.main(["--model", "block_main_b4_5", "--random-init", "--batch_size", "3", "--max_length", "40",
                     "--prompt-length", "12", "--temperature", "0.8"])
    text = capsys.readouterr().out
    assert out.shape == (3, 40)
    assert "Tok/sec" in text and "<" in text          # report + streamed token ids of the last sample


# ---- mid-batch path (17 .. 256 rows): split-K slabs + fused reduce kernels ----------------------------------------------------
def _mid_stack(layers=2, hidden=512, heads=4, inter=2048, seed=31):
    import dataclasses
    from block_transformer_amd.config import DecoderDims, ModelDims
    d = DecoderDims(hidden=hidden, heads=heads, layers=layers, inter=inter)         # head_dim 128, rot 32
    dims = ModelDims(name="mid_stack", vocab=512, block_length=4, emb_hidden=hidden // 4, expansion_ratio=2, bd=d, td=d,
                     precision="bf16")
    sd = random_state_dict(dims, seed=seed, bias_std=0.05)
    model = load_block_transformer_from_dims(dims, "fp32")
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
    return model, dims, oracle_from_dims(dims, sd16)


@pytest.mark.parametrize("B,q_len", [(32, 1), (2, 1), (300, 1), (16, 2)])
def test_decoder_forward_rejects_cache_overflow_before_any_launch(B, q_len):
    """a C caller of bt_decoder_forward with a host position past the cache end gets BT_ERR_ARG and NOTHING is
    enqueued -- on every row-count path (few-rows dual launch, 17..256-row slabs, large M), not only on the entry points the
    fused QKV + append kernels replaced.  Called through the C-ABI directly (runtime.forward has its own Python guard)."""
    import ctypes
    from block_transformer_amd import _lib
    from block_transformer_amd._lib import bt_attn_args, current_stream, dtype_code, lib, ptr
    from block_transformer_amd.runtime import KVCache, _shared_ws
    model, dims, _ = _mid_stack()
    rt = model.block_decoder._rt
    d = dims.bd
    n_max = 8
    cache = KVCache(d.layers, B, d.heads, n_max, d.hidden // d.heads, torch.bfloat16, DEV)
    cache.k.fill_(3.0)
    cache.v.fill_(3.0)
    x = torch.zeros(B * q_len, d.hidden, dtype=torch.bfloat16, device=DEV)
    out = torch.full((B * q_len, d.hidden), 7.0, dtype=torch.bfloat16, device=DEV)
    ws = _shared_ws.get(rt.workspace_bytes(B * q_len), x.device)
    rope = rt.rope(n_max)
    pos0 = n_max - q_len + 1                                   # last row would land at index n_max
    args = bt_attn_args(cache.k.data_ptr(), cache.v.data_ptr(), cache.layer_stride, cache.n_max, None, pos0, None, None, 0,
                        rope[0].data_ptr(), rope[1].data_ptr(), _lib.ATTN_ROWS, 1, None, 0)
    w = rt.weights()
    rc = lib().bt_decoder_forward(dtype_code(x.dtype), ctypes.byref(rt.dims), ctypes.byref(w), B, q_len, ptr(x),
                                  ctypes.byref(args), 0, ptr(out), ptr(ws), ws.numel(), current_stream())
    torch.cuda.synchronize()
    assert rc == -1, rc                                         # BT_ERR_ARG
    assert b"overflow" in lib().bt_last_error().lower() or b"n_max" in lib().bt_last_error().lower()
    assert bool((cache.k == 3.0).all()) and bool((cache.v == 3.0).all()) and bool((out == 7.0).all())
    # one row earlier is accepted
    args.pos0_host = pos0 - 1
    rc = lib().bt_decoder_forward(dtype_code(x.dtype), ctypes.byref(rt.dims), ctypes.byref(w), B, q_len, ptr(x),
                                  ctypes.byref(args), 0, ptr(out), ptr(ws), ws.numel(), current_stream())
    torch.cuda.synchronize()
    assert rc == 0, lib().bt_last_error()


@pytest.mark.parametrize("M", [3, 4, 6, 8, 16, 17, 32, 64, 97, 128, 200, 255, 256, 257, 400, 512])
def test_mid_rows_decoder_stack_vs_fp32_oracle(M, dbg_lib):
    """3 .. 256 rows through bt_decoder_forward (round 4: the slab path starts at 3 rows): the split-K slab GEMMs + fused reduces (bias + RoPE + KV append; bias +
    GELU; both residual projections + residual + next-layer LayerNorm pair).  A decode step (q = 1 against a 40-row cache
    built by a prefill) and a q = 2 call, hidden states and the appended K/V rows against the fp32 oracle on the same
    bf16-rounded weights and inputs (bf16 tolerance), and against the unfused launch sequence (bt_debug_mid_rows(0))."""
    model, dims, o = _mid_stack()
    H = dims.bd.hidden
    g = torch.Generator().manual_seed(M)
    past_len = 40
    xp = (torch.randn(M, past_len, H, generator=g) * 0.5).bfloat16()
    xs = (torch.randn(M, 1, H, generator=g) * 0.5).bfloat16()
    ones = torch.ones(M, past_len + 1, dtype=torch.long)

    def run():
        pre = model.block_decoder(inputs_embeds=xp.to(DEV), block_attention_mask=ones[:, :past_len].to(DEV))
        step = model.block_decoder(inputs_embeds=xs.to(DEV), block_attention_mask=ones.to(DEV),
                                   past_key_values=pre.past_key_values)
        k, v = step.past_key_values[1]                                  # layer 1 cache rows [B, heads, n, d]
        return step.hidden_states[:, 0].float().cpu(), k[:, :, past_len].float().cpu(), v[:, :, past_len].float().cpu()

    dbg_lib.bt_debug_mid_rows(1)
    hid, k_new, v_new = run()
    dbg_lib.bt_debug_mid_rows(0)
    try:
        hid_u, k_u, v_u = run()
    finally:
        dbg_lib.bt_debug_mid_rows(1)
    ref_pre, past = o.block_decoder(xp.float(), ones[:, :past_len])
    ref, kv = o.block_decoder(xs.float(), ones, past)
    ref = ref[:, 0]
    scale = ref.abs().max().item()
    assert (hid - ref).abs().max().item() <= 3e-2 * scale + 1e-2, (hid - ref).abs().max().item()
    assert (hid - hid_u).abs().max().item() <= 3e-2 * scale + 1e-2
    k_ref, v_ref = kv[1][0][:, :, past_len], kv[1][1][:, :, past_len]
    assert (k_new - k_ref).abs().max().item() <= 3e-2 * k_ref.abs().max().item() + 1e-2
    assert (v_new - v_ref).abs().max().item() <= 3e-2 * v_ref.abs().max().item() + 1e-2
    assert (k_new - k_u).abs().max().item() <= 3e-2 * k_ref.abs().max().item() + 1e-2
    if M <= 128 or M in (400, 512):   # q = 2 rows per sequence (token-decoder step 0 shape): M rows = M/2 sequences
        x2 = (torch.randn(M // 2, 2, H, generator=g) * 0.5).bfloat16()
        m2 = torch.ones(M // 2, 2, dtype=torch.long)
        got = model.block_decoder(inputs_embeds=x2.to(DEV), block_attention_mask=m2.to(DEV)).hidden_states.float().cpu()
        want, _ = o.block_decoder(x2.float(), m2)
        assert (got - want).abs().max().item() <= 3e-2 * want.abs().max().item() + 1e-2


@pytest.mark.parametrize("M", [97, 128, 129, 200, 255, 256, 257, 300, 512])
@pytest.mark.parametrize("fuse", [1, 2, 3])          # (0 = the slab + reduce launches: the reference run inside every case)
def test_mid_rows_fused_epilogue_launch_vs_oracle_and_reduce_launch(M, fuse, dbg_lib):
    """Round 6: the QKV + h->4h launch of the 97 .. 512-row path with its epilogues fused (bias, RoPE, K/V append, GELU in the GEMM's own
    accumulator layout; no reduce launch).  fuse = 1 is the product: fusion where a workgroup covers the whole K -- 257 .. 512 rows (two row
    tiles) and 129 .. 256 rows (one 128-row HALF of a column tile per workgroup, one phase per K-tile -- needs a tile count that is a multiple
    of 8, which this 512-wide stack does not have: test_mid_rows_fused_launch_at_1p2b_width_vs_oracle); 3 = fusion at 257 .. 512 rows
    only (A/B);
    2 = instead of halves, the hand-off between a tile's two K ranges up to 256 rows (measured slower, debug A/B only: one flag per tile,
    write-through slab, bounded polls); 0 = the round-5 slab + reduce launches.  Decode step (q = 1 against a 40-row cache): hidden states and the appended K / V rows vs
    the fp32 oracle, and every setting against fuse = 0 within one bf16 rounding of the same fp32 sums; repeated to catch a flag word
    left non-zero; the sticky time-out word stays clear (no BtError from the module forward)."""
    model, dims, o = _mid_stack()
    H = dims.bd.hidden
    g = torch.Generator().manual_seed(1000 + M)
    past_len = 40
    xp = (torch.randn(M, past_len, H, generator=g) * 0.5).bfloat16()
    xs = (torch.randn(M, 1, H, generator=g) * 0.5).bfloat16()
    ones = torch.ones(M, past_len + 1, dtype=torch.long)

    def run():
        pre = model.block_decoder(inputs_embeds=xp.to(DEV), block_attention_mask=ones[:, :past_len].to(DEV))
        step = model.block_decoder(inputs_embeds=xs.to(DEV), block_attention_mask=ones.to(DEV), past_key_values=pre.past_key_values)
        k, v = step.past_key_values[1]
        return step.hidden_states[:, 0].float().cpu(), k[:, :, past_len].float().cpu(), v[:, :, past_len].float().cpu()

    try:
        dbg_lib.bt_debug_mid_tall_fuse(0)
        hid0, k0, v0 = run()
        dbg_lib.bt_debug_mid_tall_fuse(fuse)
        hid, k_new, v_new = run()
        hid_b, k_b, v_b = run()
    finally:
        dbg_lib.bt_debug_mid_tall_fuse(1)
    assert torch.equal(hid, hid_b) and torch.equal(k_new, k_b) and torch.equal(v_new, v_b), "not repeatable (a hand-off flag left set?)"
    ref_pre, past = o.block_decoder(xp.float(), ones[:, :past_len])
    ref, kv = o.block_decoder(xs.float(), ones, past)
    ref = ref[:, 0]
    scale = ref.abs().max().item()
    assert (hid - ref).abs().max().item() <= 3e-2 * scale + 1e-2, (hid - ref).abs().max().item()
    k_ref, v_ref = kv[1][0][:, :, past_len], kv[1][1][:, :, past_len]
    assert (k_new - k_ref).abs().max().item() <= 3e-2 * k_ref.abs().max().item() + 1e-2
    assert (v_new - v_ref).abs().max().item() <= 3e-2 * v_ref.abs().max().item() + 1e-2
    # same fp32 sums, same formulas as the reduce launch: at most one bf16 ulp apart (multiply-add contraction may differ)
    assert (k_new - k0).abs().max().item() <= 2.0 ** -7 * k_ref.abs().max().item()
    assert (v_new - v0).abs().max().item() <= 2.0 ** -7 * v_ref.abs().max().item()
    assert (hid - hid0).abs().max().item() <= 3e-2 * scale + 1e-2


_WIDE_MID_STACK = {}


@pytest.mark.parametrize("M", [129, 200, 256])
@pytest.mark.parametrize("fuse", [1, 5])
def test_mid_rows_fused_launch_at_1p2b_width_vs_oracle(M, fuse, dbg_lib):
    """Round 6: the 129 .. 256-row forms of the fused QKV + h->4h launch need a tile count that is a multiple of 8 (the two halves of a
    column tile sit 8 block ids apart) -- the 512-wide stack of the test above never takes them.  Two layers at the 1.2B width
    (H = 2048, 16 heads, I = 8192: 112 column tiles, 32 K-tiles): fuse = 1 = the product (half mode of the slab kernel), 5 = the stream-form
    experiment of the debug build (gemm_bf16_mid_stream_kernel: W global -> registers, the k-steps of a K-tile split between a wave
    pair; measured slower, kept as an A/B).  Decode step over an 8-row cache:
    hidden states and appended K / V rows vs the fp32 oracle and vs the slab + reduce launches (fuse = 0), twice (repeatable)."""
    if "stack" not in _WIDE_MID_STACK:
        _WIDE_MID_STACK["stack"] = _mid_stack(layers=2, hidden=2048, heads=16, inter=8192, seed=33)
    model, dims, o = _WIDE_MID_STACK["stack"]
    H = dims.bd.hidden
    g = torch.Generator().manual_seed(2000 + M)
    past_len = 8
    xp = (torch.randn(M, past_len, H, generator=g) * 0.5).bfloat16()
    xs = (torch.randn(M, 1, H, generator=g) * 0.5).bfloat16()
    ones = torch.ones(M, past_len + 1, dtype=torch.long)

    def run():
        pre = model.block_decoder(inputs_embeds=xp.to(DEV), block_attention_mask=ones[:, :past_len].to(DEV))
        step = model.block_decoder(inputs_embeds=xs.to(DEV), block_attention_mask=ones.to(DEV), past_key_values=pre.past_key_values)
        k, v = step.past_key_values[1]
        return step.hidden_states[:, 0].float().cpu(), k[:, :, past_len].float().cpu(), v[:, :, past_len].float().cpu()

    try:
        dbg_lib.bt_debug_mid_tall_fuse(0)
        hid0, k0, v0 = run()
        dbg_lib.bt_debug_mid_tall_fuse(fuse)
        hid, k_new, v_new = run()
        hid_b, k_b, v_b = run()
    finally:
        dbg_lib.bt_debug_mid_tall_fuse(1)
    assert torch.equal(hid, hid_b) and torch.equal(k_new, k_b) and torch.equal(v_new, v_b), "not repeatable"
    key = ("ref", M)
    if key not in _WIDE_MID_STACK:
        _, past = o.block_decoder(xp.float(), ones[:, :past_len])
        ref, kv = o.block_decoder(xs.float(), ones, past)
        _WIDE_MID_STACK[key] = (ref[:, 0], kv[1][0][:, :, past_len], kv[1][1][:, :, past_len])
    ref, k_ref, v_ref = _WIDE_MID_STACK[key]
    scale = ref.abs().max().item()
    assert (hid - ref).abs().max().item() <= 3e-2 * scale + 1e-2, (hid - ref).abs().max().item()
    assert (k_new - k_ref).abs().max().item() <= 3e-2 * k_ref.abs().max().item() + 1e-2
    assert (v_new - v_ref).abs().max().item() <= 3e-2 * v_ref.abs().max().item() + 1e-2
    # same products, another summation order (k-steps split between two waves / one K range instead of two slabs): within one bf16 ulp
    assert (k_new - k0).abs().max().item() <= 2.0 ** -7 * k_ref.abs().max().item()
    assert (v_new - v0).abs().max().item() <= 2.0 ** -7 * v_ref.abs().max().item()
    assert (hid - hid0).abs().max().item() <= 3e-2 * scale + 1e-2


@pytest.mark.parametrize("M,mode", [(97, 1), (129, 1), (192, 1), (250, 1), (256, 1), (257, 1), (300, 1), (510, 1), (512, 1), (96, 2)])
def test_mid_rows_tall_tiles_match_the_64x64_slab_kernels(M, mode, dbg_lib):
    """97 .. 256 rows (round 5): the slab GEMMs of a layer run on 256 x 128 tiles with a K RANGE per workgroup
    (gemm_bf16_mfma256x128_slab_kernel: rows >= M clamped in the DMA addresses, unequal slices, two problems per launch, two operand
    pairs for the residual projections) instead of 64 x 64 tiles.  Same slab format, same reduce launches, another partition of K:
    hidden states and appended K/V rows of a decode step (q = 1 over a 40-row cache) and of a q = 2 call against the 64 x 64 kernels
    (bt_debug_mid_tall(0)) within fp32-summation-order noise after bf16 rounding, NaN-poisoned slab scratch (a row or column the
    kernel failed to write would surface), two other workgroup budgets (more / fewer slices) and plain instead of write-through slab
    stores (bit-identical).  mode 2 = the A/B setting that takes 65 .. 128 rows too (half of the tile clamped)."""
    from block_transformer_amd.runtime import _shared_ws
    model, dims, o = _mid_stack(layers=3)
    H = dims.bd.hidden
    g = torch.Generator().manual_seed(1000 + M)
    past_len = 40
    xp = (torch.randn(M, past_len, H, generator=g) * 0.5).bfloat16()
    xs = (torch.randn(M, 1, H, generator=g) * 0.5).bfloat16()
    ones = torch.ones(M, past_len + 1, dtype=torch.long)
    q2 = M % 2 == 0
    x2 = (torch.randn(M // 2, 2, H, generator=g) * 0.5).bfloat16()
    m2 = torch.ones(M // 2, 2, dtype=torch.long)

    def run():
        pre = model.block_decoder(inputs_embeds=xp.to(DEV), block_attention_mask=ones[:, :past_len].to(DEV))
        if _shared_ws._buf is not None:                         # fp32 NaN pattern over the whole workspace incl. the slab region
            _shared_ws._buf[:_shared_ws._buf.numel() // 4 * 4].view(torch.int32).fill_(0x7FC00000)
        step = model.block_decoder(inputs_embeds=xs.to(DEV), block_attention_mask=ones.to(DEV),
                                   past_key_values=pre.past_key_values)
        k, v = step.past_key_values[2]
        two = model.block_decoder(inputs_embeds=x2.to(DEV), block_attention_mask=m2.to(DEV)).hidden_states.float().cpu() if q2 else None
        return step.hidden_states[:, 0].float().cpu(), k[:, :, past_len].float().cpu(), v[:, :, past_len].float().cpu(), two

    outs = {}
    try:
        for key, (tall, wgs, wt) in {"small": (0, 256, 1), "tall": (mode, 256, 1), "tall512": (mode, 512, 1), "tall96": (mode, 96, 1),
                                     "plain": (mode, 256, 0)}.items():
            dbg_lib.bt_debug_mid_tall(tall)
            dbg_lib.bt_debug_mid_tall_wgs(wgs)
            dbg_lib.bt_debug_mid_tall_wgs_res(wgs)
            dbg_lib.bt_debug_mid_tall_wt(wt)
            outs[key] = run()
    finally:
        dbg_lib.bt_debug_mid_tall(1)
        dbg_lib.bt_debug_mid_tall_wgs(256)
        dbg_lib.bt_debug_mid_tall_wgs_res(256)
        dbg_lib.bt_debug_mid_tall_wt(1)
    ref_pre, past = o.block_decoder(xp.float(), ones[:, :past_len])
    ref, kv = o.block_decoder(xs.float(), ones, past)
    ref = ref[:, 0]
    scale = ref.abs().max().item()
    base = outs["small"]
    for a_, b_ in zip(outs["plain"], outs["tall"]):               # the store policy of the slabs changes no value
        assert a_ is None or torch.equal(a_, b_)
    for key in ("tall", "tall512", "tall96"):
        got = outs[key]
        assert torch.isfinite(got[0]).all() and torch.isfinite(got[1]).all() and torch.isfinite(got[2]).all(), key
        assert (got[0] - ref).abs().max().item() <= 3e-2 * scale + 1e-2, (key, (got[0] - ref).abs().max().item())
        # against the 64 x 64 kernels (above 256 rows: against the large-M launch sequence, which the stack takes with the knob off):
        # three layers of bf16 activations whose fp32 pre-rounding sums differ in the last bits
        tol = 1.5e-2 if M <= 256 else 3e-2
        assert (got[0] - base[0]).abs().max().item() <= tol * scale, (key, (got[0] - base[0]).abs().max().item())
        assert (got[0] - base[0]).abs().mean().item() <= tol / 15 * scale, (key, (got[0] - base[0]).abs().mean().item())
        for i in (1, 2):
            assert (got[i] - base[i]).abs().max().item() <= tol * base[i].abs().max().item(), (key, i)
        if q2:
            assert torch.isfinite(got[3]).all()
            assert (got[3] - base[3]).abs().max().item() <= tol * base[3].abs().max().item(), key
    if q2:
        want, _ = o.block_decoder(x2.float(), m2)
        assert (outs["tall"][3] - want).abs().max().item() <= 3e-2 * want.abs().max().item() + 1e-2


@pytest.mark.parametrize("B", [24, 64, 200])
def test_mid_batch_generate_matches_unfused(B, dbg_lib):
    """generate() at 17 .. 256 sequences on the 2-layer 1.2B-shaped model: the slab + fused-reduce stacks against the unfused
    launch sequence (same weights, same prompt): first token of every sequence equal except rare bf16 near-ties, context
    of agreeing sequences within bf16 rounding; and the first block against the fp32 oracle for 3 sequences."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=5, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(0, dims.vocab, (B, 32), generator=torch.Generator().manual_seed(B)).to(DEV)
    dbg_lib.bt_debug_mid_rows(1)
    fused = model.generate(prompt, max_length=48).cpu()
    dbg_lib.bt_debug_mid_rows(0)
    try:
        model._engine = None
        unfused = model.generate(prompt, max_length=48).cpu()
    finally:
        dbg_lib.bt_debug_mid_rows(1)
        model._engine = None
    assert torch.equal(fused[:, :32], prompt.cpu())
    assert (fused[:, 32] == unfused[:, 32]).float().mean() >= 0.9, (fused[:, 32] == unfused[:, 32]).float().mean()
    # margin-aware (8 rows spread over the batch, both launch sequences): every token of the first block is the fp32 oracle's
    # argmax or inside the bf16 noise band below it
    sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
    o = oracle_from_dims(dims, sd16)
    rows = torch.linspace(0, B - 1, 8).long()
    for name, out in (("fused", fused), ("unfused", unfused)):
        _, ref_lg = oracle_block_logits(o, prompt.cpu()[rows], out[rows, 32:36])
        assert_tokens_within_noise(ref_lg, out[rows, 32:36], 3e-2, 1e-2, f"mid-batch B={B} {name}")
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("M", [24, 64, 200])
def test_mid_rows_in_launch_reduction_is_exact_under_load(M, dbg_lib):
    """The in-launch split-K reduction (last-arriving slice sums the fp32 slabs; agent-scope release / acquire hand-off across
    CUs and XCDs) against the separate reduce kernels (bt_debug_mid_fuse(0)): same slabs, same summation order -> outputs
    agree to one bf16 ulp on EVERY element, 40 times in a row, consumer L1-warm (same addresses every pass) and while a
    second stream keeps the memory system busy (a stale or torn slab read would be off by a whole partial sum)."""
    model, dims, o = _mid_stack(layers=3)
    H = dims.bd.hidden
    g = torch.Generator().manual_seed(100 + M)
    x = (torch.randn(M, 1, H, generator=g) * 0.5).bfloat16().to(DEV)
    ones = torch.ones(M, 1, dtype=torch.long, device=DEV)
    side = torch.cuda.Stream()
    junk_a = torch.empty(64 << 20, dtype=torch.float32, device=DEV).normal_()
    junk_b = torch.empty_like(junk_a)

    def run():
        return model.block_decoder(inputs_embeds=x, block_attention_mask=ones).hidden_states.float()

    dbg_lib.bt_debug_mid_fuse(0)
    ref = run()
    dbg_lib.bt_debug_mid_fuse(2)       # in-launch reduction for every slice count (A/B mode; the default fuses one-slice GEMMs only)
    tol = ref.abs() * 2.0 ** -7 + 1e-3
    try:
      for it in range(40):
        if it % 2 == 0:
            with torch.cuda.stream(side):                      # uneven background load on every other pass
                for _ in range(4):
                    junk_b.copy_(junk_a)
        got = run()
        bad = (got - ref).abs() > tol
        assert not bool(bad.any()), (it, int(bad.sum()), float((got - ref).abs().max()))
    finally:
        dbg_lib.bt_debug_mid_fuse(1)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B", [1, 2, 4, 6, 16])
def test_few_rows_dual_launch_is_bit_identical(B, dbg_lib):
    """<= 16 rows (GEMV up to 4, MFMA weight streamer above): QKV (LN1 / RoPE / append) and h->4h (LN2 / GELU) of a layer in ONE launch (gemv_qkv_h4_kernel) == the two separate
    GEMV launches, bit for bit (same per-row arithmetic, only the launch grouping changes): tokens and last context equal."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=5, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(0, dims.vocab, (B, 64), generator=torch.Generator().manual_seed(B)).to(DEV)
    dbg_lib.bt_debug_few_dual(1)
    a = model.generate(prompt, max_length=96).cpu()
    ctx_a = model._engine.ctx.float().cpu().clone()
    dbg_lib.bt_debug_few_dual(0)
    try:
        model._engine = None
        b = model.generate(prompt, max_length=96).cpu()
        ctx_b = model._engine.ctx.float().cpu().clone()
    finally:
        dbg_lib.bt_debug_few_dual(1)
        model._engine = None
    assert torch.equal(a, b) and torch.equal(ctx_a, ctx_b)
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B,L", [(1, 4), (2, 8)])
def test_few_rows_fused_attention_is_bit_identical(B, L, dbg_lib):
    """Round 4: <= 4 rows, token decoder: the local-window attention is computed INSIDE the residual-stream GEMV (gemv_attn_res_kernel:
    every workgroup recomputes the <= 12-key window of all heads with local_attend_kernel's lane mapping and arithmetic), so the
    layer is 2 launches.  == attention launch + bt_linear2, bit for bit: tokens and the last context embedding equal, graphs on
    (what the engine runs) and off.  L = 4 -> window <= 5 keys (U = 2), L = 8 -> <= 9 keys (U = 3); B = 1 / 2 / 4 -> step 0 has
    M = 2 / 4 / 8 rows (8 rows = MFMA streamer, not this path), steps 1.. have M = 1 / 2 / 4."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b" if L == 4 else "block_main_b8_1.2b")
    dims = dataclasses.replace(base, name=f"b{L}_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=5, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(0, dims.vocab, (B, 64), generator=torch.Generator().manual_seed(B)).to(DEV)
    outs = {}
    try:
        for fuse in (1, 0):
            dbg_lib.bt_debug_few_attn(fuse)
            for graphs in (True, False):
                model._engine = None
                model.use_graphs = graphs
                out = model.generate(prompt, max_length=64 + 5 * L).cpu()
                outs[(fuse, graphs)] = (out, model._engine.ctx.float().cpu().clone())
    finally:
        dbg_lib.bt_debug_few_attn(1)
        model.use_graphs = True
        model._engine = None
    ref_out, ref_ctx = outs[(0, False)]
    for key, (out, ctx) in outs.items():
        assert torch.equal(out, ref_out) and torch.equal(ctx, ref_ctx), key
    assert len({tuple(r.tolist()) for r in ref_out[:, 64:]}) == B or B == 1
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B,L", [(24, 4), (64, 4), (256, 4), (32, 8)])
def test_mid_batch_dual_gemm_matches_separate_launches(B, L, dbg_lib):
    """Round 4: 17..64 rows: the layer's QKV projection and MLP up-projection as ONE split-K launch (gemm_bf16_mfma_dual_kernel), their
    reductions as one launch, and -- token decoder -- the local-window attention inside that reduce launch
    (mid_reduce_qkv_attn_gelu_kernel: 4 launches per layer), against (a) the same without the attention fold and (b) the two
    launches each of round 2 (bt_debug_mid_dual(0)).  The K slicing may differ (224 instead of 96 / 128 tiles per row block) and the
    folded softmax runs sequentially instead of over four lane groups, so equality is up to fp32 summation order: first tokens
    equal except rare bf16 near-ties; every variant against the fp32 oracle, margin-aware; graphs on == off for the production
    path, bit for bit.  B = 256 (round 5): the shared launch runs on 256 x 128 tiles with two K ranges (gemm_bf16_mfma256x128_slab_kernel)
    and the attention rides in its reduce launch -- same comparison, same tolerances; with bt_debug_mid_tall(0) one K slice fills the
    chip, the separate fused-epilogue launches stay and all variants are equal bit for bit.  L = 8: windows of up to 9 keys (KMAX = 12
    instantiation)."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b" if L == 4 else "block_main_b8_1.2b")
    dims = dataclasses.replace(base, name=f"b{L}_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=5, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    P = 8 * L
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(B)).to(DEV)
    outs = {}
    try:
        for name, dual, attn in (("dual+attn", 1, 1), ("dual", 1, 0), ("separate", 0, 0)):
            dbg_lib.bt_debug_mid_dual(dual)
            dbg_lib.bt_debug_mid_attn(attn)
            model._engine = None
            outs[name] = model.generate(prompt, max_length=P + 4 * L).cpu()
        dbg_lib.bt_debug_mid_dual(1)
        dbg_lib.bt_debug_mid_attn(1)
        model._engine = None
        model.use_graphs = False
        eager = model.generate(prompt, max_length=P + 4 * L).cpu()
    finally:
        dbg_lib.bt_debug_mid_dual(1)
        dbg_lib.bt_debug_mid_attn(1)
        model.use_graphs = True
        model._engine = None
    assert torch.equal(outs["dual+attn"], eager)
    assert torch.equal(outs["dual+attn"][:, :P], prompt.cpu())
    if B > 224:   # the 64 x 64 kernels at > 224 rows: one K slice, fused epilogues, no shared launch -> the knobs change nothing
        try:
            dbg_lib.bt_debug_mid_tall(0)
            small = {}
            for name, dual in (("dual", 1), ("separate", 0)):
                dbg_lib.bt_debug_mid_dual(dual)
                model._engine = None
                small[name] = model.generate(prompt, max_length=P + 4 * L).cpu()
        finally:
            dbg_lib.bt_debug_mid_tall(1)
            dbg_lib.bt_debug_mid_dual(1)
            model._engine = None
        assert torch.equal(small["dual"], small["separate"])
    for name in ("dual+attn", "dual"):
        agree = (outs[name][:, P] == outs["separate"][:, P]).float().mean()
        assert agree >= 0.9, (name, agree)
    sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
    o = oracle_from_dims(dims, sd16)
    rows = torch.linspace(0, B - 1, 8).long()
    for name, out in outs.items():
        _, ref_lg = oracle_block_logits(o, prompt.cpu()[rows], out[rows, P:P + L])
        assert_tokens_within_noise(ref_lg, out[rows, P:P + L], 3e-2, 1e-2, f"mid-batch B={B} L={L} {name}")
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B", [1, 5, 64, 300, 600])
def test_unwritten_cache_rows_never_enter_the_attention(B):
    """Round 4 (found by test_greedy_tokens_do_not_depend_on_keeping_logits on the folded mid-batch attention): a masked or not yet
    written cache row gets softmax weight exactly 0, but 0 x NaN is NaN -- every attention path must keep such rows out of its sums
    as DATA, not only as weights.  The engine's global and local KV caches are poisoned with NaN between two generate() calls of
    the same prompt (prefill and decode rewrite exactly the rows they own): identical tokens, all batch-size paths (GEMV + local
    kernel, slab path with the folded attention -- 64 x 64 and, B = 300, 256 x 128 tiles --, large M with flash prefill), graphs on and off."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=6, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(0, dims.vocab, (B, 30), generator=torch.Generator().manual_seed(B)).to(DEV)   # P % L != 0: left-padded block
    try:
        for graphs in (True, False):
            model.use_graphs = graphs
            model._engine = None
            want = model.generate(prompt, max_length=50).cpu()
            eng = model._engine
            with torch.inference_mode():
                for t_ in (eng.glob.k, eng.glob.v, eng.tok.local.k, eng.tok.local.v):
                    t_.fill_(float("nan"))
            got = model.generate(prompt, max_length=50).cpu()
            assert torch.equal(got, want), (graphs, (got != want).nonzero()[:4])
            assert int(got.min()) >= 0 and int(got.max()) < dims.vocab
    finally:
        model.use_graphs = True
        model._engine = None
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B", [600])
def test_two_branch_parallel_residual_is_bit_identical(B, dbg_lib):
    """(Debug build only since round 5: measured -0.2 ... -0.3 % end to end, the product library ignores the second stream.)
    Round 4: with a second stream (bt_attn_args.aux_stream; model.two_branch = True) stacks of > 256 bf16 rows run
    mlp.dense_h_to_4h beside QKV -> attention -> attention.dense (GPT-NeoX parallel residual: it only reads LN2 of the layer input)
    and join in front of dense_4h_to_h.  Same kernels, same arithmetic, same order of the residual updates: tokens and the last
    context embedding are bit-identical to the one-stream schedule, captured as a forked hipGraph and launched eagerly."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=7, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(0, dims.vocab, (B, 64), generator=torch.Generator().manual_seed(B)).to(DEV)
    outs = {}
    try:
        for two in (True, False):
            for graphs in (True, False):
                model.two_branch, model.use_graphs = two, graphs
                model._engine = None
                out = model.generate(prompt, max_length=64 + 12).cpu()
                assert (model._engine.aux is not None) == two
                outs[(two, graphs)] = (out, model._engine.ctx.float().cpu().clone())
    finally:
        model.two_branch, model.use_graphs = None, True
        model._engine = None
    ref_out, ref_ctx = outs[(False, False)]
    for key, (out, ctx) in outs.items():
        assert torch.equal(out, ref_out) and torch.equal(ctx, ref_ctx), key
    del model
    torch.cuda.empty_cache()


def test_layernorm_folded_into_the_gemms_matches_the_layernorm_launch(dbg_lib):
    """Round 5 EXPERIMENT (debug library only: measured no net gain, profiles/r5_ab_ln_fused.jsonl): above 256 rows the dual LayerNorm
    launch of a layer is replaced by row statistics + the folded form inside the QKV and h->4h GEMMs (bt_layer_weights.*_wf; raw
    residual stream x folded weight, epilogue rstd (acc - mean s) + c; DecoderRuntime.fold_layernorm = True).  2-layer 1.2B-shaped
    bf16 model with non-trivial LayerNorm weights / biases, B = 4096 (prefill chunks of 32768 rows, token step 0 at 8192 rows, the
    other steps and the block-decoder step at 4096 rows all take the folded path): context embedding of the second block and every
    token of two blocks against the fp32 oracle on the same bf16 weights, folded and unfolded (debug knob), and the two against each
    other within the same budget."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=13, include_unused_embed_in=False, bias_std=0.05)     # LayerNorm gamma = 1 + N(0, .05), beta = N(0, .05)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    model.keep_logits = True
    o = oracle_from_dims(dims, {k: v.bfloat16().float() for k, v in sd.items()})
    B, P, L = 4096, 32, dims.L
    prompt = torch.randint(0, dims.vocab, (B, P), generator=torch.Generator().manual_seed(5))
    rows = [0, 1999, B - 1]
    res = {}
    for rt in (model.block_decoder._rt, model.token_decoder._rt):
        rt.fold_layernorm = True
    try:
        for fold in (1, 0):
            lib_ = dbg_lib
            lib_.bt_debug_ln_fold(fold)
            model._engine = None
            out = model.generate(prompt.to(DEV), max_length=P + L + 1).cpu()
            eng = model._engine
            res[fold] = (out, eng.ctx.float().cpu()[rows], eng.tok.logits.float().cpu()[rows])
    finally:
        dbg_lib.bt_debug_ln_fold(1)
        model._engine = None
    assert model.block_decoder._rt._folded is not None and len(model.block_decoder._rt._folded) == 2
    ref_ctx2, ref_lg2 = None, None
    for fold, (out, ctx2, lg2) in res.items():
        ref_ctx2, ref_lg2 = _oracle_first_token(o, out[rows, :P + L])
        e_ctx, e_lg = (ctx2 - ref_ctx2).abs().max().item(), (lg2 - ref_lg2).abs().max().item()
        b_ctx, b_lg = 3e-2 * ref_ctx2.abs().max().item() + 1e-2, 3e-2 * ref_lg2.abs().max().item() + 1e-2
        print(f"[ln fold={fold}] block-2 ctx err {e_ctx:.4f} / {b_ctx:.4f}, logit err {e_lg:.4f} / {b_lg:.4f}")
        assert e_ctx <= b_ctx and e_lg <= b_lg
        _, ref_blk1 = oracle_block_logits(o, prompt[rows], out[rows, P:P + L])
        assert_tokens_within_noise(ref_blk1, out[rows, P:P + L], 3e-2, 1e-2, f"ln fold={fold} first block")
        assert torch.equal(out[:, :P], prompt)
    d_ctx = (res[1][1] - res[0][1]).abs().max().item()
    assert d_ctx <= 3e-2 * res[0][1].abs().max().item() + 1e-2, d_ctx
    assert (res[1][0][:, P] == res[0][0][:, P]).float().mean().item() >= 0.97     # first tokens: bf16 near-ties only
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B", [1, 2])
def test_few_rows_persistent_gemv_is_bit_identical(B, dbg_lib):
    """Round 4: <= 2 rows: the [QKV | h->4h] launch as ~2 resident workgroups per CU that stage and LayerNorm x once and walk their
    row groups with the next group's weight rows in flight (gemv_qkv_h4_persist_kernel) == one workgroup per row group
    (gemv_qkv_h4_kernel), bit for bit -- same piece ownership, LayerNorm, dot / butterfly order and epilogues: tokens and the last
    context embedding equal, graphs on and off.  B = 1: M = 1 (steps 1..) and 2 (step 0); B = 2: M = 2 and (step 0) 4 rows -> slab path."""
    import dataclasses
    base = named_dims("block_main_b4_1.2b")
    dims = dataclasses.replace(base, name="b4_1.2b_2layer", bd=dataclasses.replace(base.bd, layers=2),
                               td=dataclasses.replace(base.td, layers=2))
    sd = random_state_dict(dims, seed=5, include_unused_embed_in=False)
    model = load_block_transformer_from_dims(dims, "fp32", keep_unused_embed_in=False)
    model.load_state_dict(sd, strict=True)
    model = model.bfloat16().to(DEV)
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(0, dims.vocab, (B, 64), generator=torch.Generator().manual_seed(B)).to(DEV)
    outs = {}
    try:
        for persist in (1, 0):
            dbg_lib.bt_debug_few_persist(persist)
            for graphs in (True, False):
                model._engine = None
                model.use_graphs = graphs
                out = model.generate(prompt, max_length=96).cpu()
                outs[(persist, graphs)] = (out, model._engine.ctx.float().cpu().clone())
    finally:
        dbg_lib.bt_debug_few_persist(1)
        model.use_graphs = True
        model._engine = None
    ref_out, ref_ctx = outs[(0, False)]
    for key, (out, ctx) in outs.items():
        assert torch.equal(out, ref_out) and torch.equal(ctx, ref_ctx), key
    del model
    torch.cuda.empty_cache()
