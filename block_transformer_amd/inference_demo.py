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
    if isinstance(model, BlockTransformer):
        model.token_decoder.generation_config.update(do_sample=True, temperature=temperature)
    else:
        model.generation_config.update(do_sample=True, temperature=temperature)


class FirstSampleStreamer:
    """Prints the tokens of the last sample as they are generated (reference :69-83)."""

    def __init__(self, tokenizer=None):
        self.tokenizer = tokenizer
        self.first = True

    def put(self, value):
        if self.first:  # ignore the prompt / placeholder
            self.first = False
            return
        tok = value[-1]
        text = self.tokenizer.decode(tok).replace("\n", "\\n") if self.tokenizer is not None else f"<{int(tok)}>"
        print(text, end="", flush=True)

    def end(self):
        self.first = True


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
    args = ap.parse_args(argv)

    model, tokenizer = load_model(args.model, args.checkpoint_root, args.random_init, args.tokenizer)
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
    streamer = None if args.no_stream else FirstSampleStreamer(tokenizer)
    torch.cuda.synchronize()
    start = time.time()
    output_ids = model.generate(**inputs, max_length=args.max_length, streamer=streamer)
    torch.cuda.synchronize()
    duration = time.time() - start
    memory = torch.cuda.max_memory_allocated(device=None)
    print("\n" + " End of generation ".center(60, "-"))
    print(f"Prompt length              : {prompt_length:>8}")
    print(f"Generation length          : {generation_length:>8}")
    print(f"Batch size                 : {args.batch_size:>8}")
    print(f"Max memory allocated       : {memory / 1024 ** 3:>6.2f}GB")
    print(f"Tok/sec/sample             : {generation_length / duration:>8.2f}")
    print(f"Tok/sec                    : {generation_length * args.batch_size / duration:>8.2f}")
    return output_ids


if __name__ == "__main__":
    main()
