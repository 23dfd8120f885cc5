"""Demo harness — the MI355X counterpart of the reference's inference_demo.py (same flow, same knobs):

load the newest ``checkpoint-N/model.safetensors`` under ``<checkpoint_root>/<model>`` by the reference's state-dict key
names (:24-59) -> ``set_temperature`` (do_sample + temperature on ``token_decoder.generation_config``, HF default
top_k=50, :62-66) -> tokenize the prompt x batch (or use ``--prompt-ids`` / random ids when no local tokenizer exists;
the hub is unreachable here) -> ``model.generate(..., streamer=FirstSampleStreamer)`` -> tokens/s and peak memory
(:143-161).

    python -m block_transformer_amd.inference_demo --model block_main_b4_1.2b --batch_size 128 --max_length 1334 \
        [--checkpoint_root results] [--tokenizer /path/to/pythia-tokenizer] [--random-init]
"""
from __future__ import annotations

import argparse
import os
import re
import time
from typing import Optional

import torch

from .config import load_config
from .model.block_transformer import BlockTransformer
from .model.utils import load_block_transformer_from_config


def get_checkpoint_path(name: str, checkpoint_root: str) -> str:
    """Newest ``checkpoint-<step>`` directory (reference :24-41)."""
    root = os.path.join(checkpoint_root, name)
    steps = []
    for d in os.listdir(root) if os.path.isdir(root) else []:
        m = re.fullmatch(r"checkpoint-(\d+)", d)
        if m and os.path.exists(os.path.join(root, d, "model.safetensors")):
            steps.append(int(m.group(1)))
    if not steps:
        raise FileNotFoundError(f"no checkpoint-*/model.safetensors under {root}")
    return os.path.join(root, f"checkpoint-{max(steps)}", "model.safetensors")


def load_model(name: str, checkpoint_root: Optional[str], random_init: bool = False, tokenizer_path=None,
               device="cuda"):
    cfg = load_config(name)
    if tokenizer_path:
        cfg.setdefault("tokenizer", {})["local_path"] = tokenizer_path
    with torch.device(device):
        model, tokenizer = load_block_transformer_from_config(cfg)
    if not random_init:
        from safetensors.torch import load_file
        sd = load_file(get_checkpoint_path(name, checkpoint_root), device=str(device))
        missing, unexpected = model.load_state_dict(sd, strict=False)
        # the reference's unused block_decoder.gpt_neox.embed_in may be absent in slimmed checkpoints (Q8)
        missing = [k for k in missing if k != "block_decoder.gpt_neox.embed_in.weight"]
        if missing or unexpected:
            raise RuntimeError(f"checkpoint does not match the reference key layout: missing={missing[:5]} "
                               f"unexpected={unexpected[:5]}")
    return model.eval(), tokenizer


def set_temperature(model, temperature: float):
    """Switch generation to sampling at ``temperature`` (HF default top_k = 50) — reference inference_demo.py:62-66."""
    holder = model.token_decoder if hasattr(model, "token_decoder") else model
    holder.generation_config.update(do_sample=True, temperature=float(temperature))


class SampleStreamer:
    """HF-streamer-protocol sink (``put(tensor)`` per step, ``end()`` per generate call) that echoes ONE sample of the batch
    as it is generated.  ``BlockTransformer.generate`` drives it once per block: a ``[B, 1]`` placeholder first (dropped
    here, like the prompt in HF's TextStreamer), then one ``[B]`` tensor per token, then ``end()``."""

    def __init__(self, tokenizer=None, sample_index: int = -1):
        self.tokenizer = tokenizer
        self.sample_index = sample_index
        self._drop_next = True

    def _render(self, token_id: int) -> str:
        if self.tokenizer is None:
            return f"<{token_id}>"
        return self.tokenizer.decode(token_id).replace("\n", "\\n")

    def put(self, value):
        if self._drop_next:
            self._drop_next = False
            return
        print(self._render(int(value.reshape(-1)[self.sample_index])), end="", flush=True)

    def end(self):
        self._drop_next = True


FirstSampleStreamer = SampleStreamer   # the reference's name for it (inference_demo.py:69-83)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="block_main_b4_1.2b")
    ap.add_argument("--checkpoint_root", default="results")
    ap.add_argument("--batch_size", type=int, default=128)
    ap.add_argument("--max_length", type=int, default=1334)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--tokenizer", default=None, help="local tokenizer directory (no hub access here)")
    ap.add_argument("--prompt", default="It was the best of times, it was the worst of times,")
    ap.add_argument("--prompt-length", type=int, default=64, help="random prompt length when no tokenizer is given")
    ap.add_argument("--random-init", action="store_true", help="skip the checkpoint (plumbing / speed demo)")
    ap.add_argument("--no-stream", action="store_true")
    ap.add_argument("--rope-policy", choices=["hf439_generate", "train"], default=None,
                    help="token-decoder RoPE positions of cached generation (default: the reference's, hf439_generate)")
    args = ap.parse_args(argv)

    model, tokenizer = load_model(args.model, args.checkpoint_root, args.random_init, args.tokenizer)
    if args.rope_policy:
        model.token_decoder.rope_policy = args.rope_policy
    if tokenizer is not None:
        inputs = tokenizer([args.prompt] * args.batch_size, return_tensors="pt")
        inputs = {k: t.cuda() for k, t in inputs.items() if k in ("input_ids", "attention_mask")}
    else:
        ids = torch.randint(2, model.config.vocab_size, (1, args.prompt_length),
                            generator=torch.Generator().manual_seed(0)).repeat(args.batch_size, 1)
        inputs = {"input_ids": ids.cuda()}
    prompt_length = inputs["input_ids"].shape[1]
    generation_length = args.max_length - prompt_length
    set_temperature(model, args.temperature)
    streamer = None if args.no_stream else SampleStreamer(tokenizer)
    torch.cuda.synchronize()
    start = time.time()
    output_ids = model.generate(**inputs, max_length=args.max_length, streamer=streamer)
    torch.cuda.synchronize()
    duration = time.time() - start
    report = [("Prompt length", f"{prompt_length}"), ("Generation length", f"{generation_length}"),
              ("Batch size", f"{args.batch_size}"),
              ("Max memory allocated", f"{torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB"),
              ("Tok/sec/sample", f"{generation_length / duration:.2f}"),
              ("Tok/sec", f"{generation_length * args.batch_size / duration:.2f}")]
    print("\n" + "=" * 48)
    for label, value in report:
        print(f"{label:<24}{value:>24}")
    return output_ids


if __name__ == "__main__":
    main()
