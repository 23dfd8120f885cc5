"""Throughput harness — the MI355X counterpart of the reference's measure_generation_time.py, same protocol:

* random-init model from the YAML dims, EOS disabled, greedy (reference :62-92);
* ``max_length = ceil(P/L)*L + D``; block-decoder positions ``ceil(max_length/L) - 1`` (:44-49,65-70);
* inputs ``randint(0, vocab, (B, P))``; device events around the whole ``generate()`` (:131-161);
* 1 warm-up + 5 timed runs (10 if max_length < 256), mean/std, per-sample and per-token times, peak memory (:179-233);
* OOM -> recorded, not fatal (:237-249); batch auto-tune: measure B=1,2 -> linear memory model -> bisect between the
  largest OK batch and the predicted / smallest-OOM batch, rounded to multiples of 4/16/64/256, stop below 2 GiB free
  (:319-372,445-468);
* CSV ``results/<name>/generation_time.csv`` with the reference's columns (:217-233,488-491) plus tokens/s.

    python -m block_transformer_amd.measure_generation_time --config-name block_main_b4_1.2b \
        ++benchmark_prefill_length=2048 ++benchmark_decode_length=128 [++benchmark_batch_sizes=[64,256,1024]]
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time
from typing import Dict, List, Optional

import numpy as np
import torch

from .config import apply_overrides, dims_from_cfg, get_max_length, load_config
from .model.block_transformer import BlockTransformer
from .model.utils import load_block_transformer_from_dims

DEVICE = "cuda"
DEFAULT_PREFILL_LENGTH = 1
DEFAULT_DECODE_LENGTH = 2048
SAVE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "results")


def get_number_of_repetitions(max_length: int) -> int:
    return 10 if max_length < 256 else 5


def prepare_block_model(cfg, prefill_length: int, decode_length: int) -> BlockTransformer:
    dims = dims_from_cfg(cfg)
    with torch.device(DEVICE):
        model = load_block_transformer_from_dims(dims, cfg.get("precision", "bf16"), keep_unused_embed_in=False)
    max_length = get_max_length(dims.block_length, prefill_length, decode_length)
    model.block_decoder.config.max_position_embeddings = math.ceil(max_length / dims.block_length) - 1
    model.token_decoder.config.pad_token_id = 1
    model.token_decoder.config.eos_token_id = 0
    model.token_decoder.generation_config.eos_token_id = None
    policy = (cfg.get("token_decoder") or {}).get("rope_policy")       # build-specific knob, see engine.td_position_table
    if policy is not None:
        model.token_decoder.rope_policy = policy
    return model.eval()


def generate_with_block_model_and_measure_time(model: BlockTransformer, prefill_length, decode_length, batch_size,
                                               log_path=None) -> float:
    """milliseconds for one generate() (H2D copy of the ids included, like the reference)."""
    max_length = get_max_length(model.block_length, prefill_length, decode_length)
    input_ids = torch.randint(0, model.config.vocab_size, (batch_size, prefill_length))
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    if log_path is None:
        output = model.generate(input_ids.to(DEVICE), max_length=max_length, benchmark=False)
    else:
        output = _generate_profiled(model, input_ids.to(DEVICE), max_length, log_path)
    end.record()
    assert output.shape[1] == max_length, output.shape
    torch.cuda.synchronize()
    return start.elapsed_time(end)


def _generate_profiled(model: BlockTransformer, input_ids, max_length: int, log_path: str):
    """The reference wraps this generate() in torch.profiler and exports a Chrome trace (measure_generation_time.py:150-155).
    The kernels here are launched by libbt_hip, so the trace comes from its own timer: the generate runs eagerly (no graph
    replay -- a replayed graph is one opaque launch) with a hipEvent pair around every kernel, the ten most expensive
    kernel families are printed and every launch becomes one complete ("X") event of a chrome://tracing / Perfetto file."""
    import json
    from . import _lib
    graphs = model.use_graphs
    model.use_graphs = False
    _lib.prof_enable(True)
    try:
        output = model.generate(input_ids, max_length=max_length, benchmark=False)
        torch.cuda.synchronize()
        launches = _lib.prof_trace()
        families = _lib.prof_collect()
    finally:
        _lib.prof_enable(False)
        model.use_graphs = graphs
    total = sum(ms for ms, _ in families.values()) or 1.0
    print(f"{'kernel family':<18}{'GPU ms':>12}{'%':>8}{'launches':>10}{'avg us':>10}")
    for name, (ms, n) in sorted(families.items(), key=lambda kv: -kv[1][0])[:10]:
        if n:
            print(f"{name:<18}{ms:>12.3f}{100 * ms / total:>8.2f}{n:>10d}{1e3 * ms / n:>10.1f}")
    events = [{"name": fam, "cat": "kernel", "ph": "X", "pid": 0, "tid": 0, "ts": 1e3 * t0, "dur": 1e3 * d}
              for fam, t0, d in launches]
    events.append({"name": "process_name", "ph": "M", "pid": 0, "args": {"name": "libbt_hip launch stream (eager)"}})
    with open(log_path, "w") as f:
        json.dump({"traceEvents": events, "displayTimeUnit": "ms"}, f)
    print(f"trace of {len(launches)} launches written to {log_path}")
    return output


def measure_generation(model, prefill_length, decode_length, batch_size, log_path=None) -> Dict:
    print(f" ( prefill={prefill_length} decode={decode_length} / bs={batch_size} ) ".center(80, "-"))
    # the engine of the previous batch size holds its KV caches: drop it and hand the blocks back to the driver first (the
    # reference's model keeps no state between calls, so its empty_cache() alone does this)
    model.release_buffers()
    torch.cuda.empty_cache()
    try:
        print("Warming up... ", end="", flush=True)
        t = generate_with_block_model_and_measure_time(model, prefill_length, decode_length, batch_size)
        print(f"done in {t:.2f}ms")
        torch.cuda.reset_peak_memory_stats(DEVICE)
        repeat = get_number_of_repetitions(get_max_length(model.block_length, prefill_length, decode_length))
        times = np.array([generate_with_block_model_and_measure_time(model, prefill_length, decode_length, batch_size,
                                                                     log_path) for _ in range(repeat)])
        gib = 1024 ** 3
        mean = float(times.mean())
        stats = {
            "prefill_length": prefill_length, "decode_length": decode_length, "batch_size": batch_size, "oom": False,
            "mean": mean, "mean_per_sample": mean / batch_size, "mean_per_token": mean / decode_length / batch_size,
            "max_memory_allocated": torch.cuda.max_memory_allocated(DEVICE) / gib,
            "max_memory_reserved": torch.cuda.max_memory_reserved(DEVICE) / gib,
            "tokens_per_s": batch_size * decode_length / (mean / 1e3),
        }
        if len(times) > 1:
            stats["std"] = float(times.std())
        print(f"mean:                 {mean:8.2f}ms / {stats['mean_per_sample']:8.2f}ms per sample / "
              f"{stats['mean_per_token']:8.4f}ms per token  ({stats['tokens_per_s']:.0f} tok/s)")
        print(f"max memory_allocated: {stats['max_memory_allocated']:8.2f}GB")
        return stats
    except RuntimeError as e:
        if isinstance(e, torch.cuda.OutOfMemoryError) or "out of memory" in str(e):
            print(f"\nOut of memory using batch size {batch_size} for prefill={prefill_length} decode={decode_length}")
            model.release_buffers()
            torch.cuda.empty_cache()
            return {"batch_size": batch_size, "prefill_length": prefill_length, "decode_length": decode_length,
                    "oom": True}
        raise


# ---- batch-size search (pure bookkeeping over the measurement records; unit-tested on CPU) ------------------------------
def _rows(data: List[Dict], prefill_length, decode_length, oom: Optional[bool] = None) -> List[Dict]:
    rows = [r for r in data if r["prefill_length"] == prefill_length and r["decode_length"] == decode_length]
    if oom is not None:
        rows = [r for r in rows if bool(r["oom"]) == oom]
    return sorted(rows, key=lambda r: r["batch_size"])


def predict_oom(data, batch_size, prefill_length, decode_length) -> bool:
    return any(r["batch_size"] <= batch_size for r in _rows(data, prefill_length, decode_length, oom=True))


def measurement_exists(data, batch_size, prefill_length, decode_length) -> bool:
    return any(r["batch_size"] == batch_size for r in _rows(data, prefill_length, decode_length))


def get_max_batch_size_row(data, prefill_length, decode_length, oom=False) -> Optional[Dict]:
    rows = _rows(data, prefill_length, decode_length, oom=oom)
    return rows[-1] if rows else None


def predict_memory_per_sample(data, prefill_length, decode_length) -> Optional[float]:
    ok = _rows(data, prefill_length, decode_length, oom=False)
    if len(ok) < 2:
        print("Not enough data to predict memory per sample.")
        return None
    b1, b2 = ok[0]["batch_size"], ok[-1]["batch_size"]
    m1, m2 = ok[0]["max_memory_allocated"], ok[-1]["max_memory_allocated"]
    per = (m2 - m1) / (b2 - b1)
    print(f"Predicted memory per sample: {per} GiB")
    if per <= 0:
        if b1 == 1 and b2 == 2:
            per = 0.1  # the usage is too small to resolve: assume 0.1 GiB like the reference
        else:
            raise ValueError("Predicted memory per sample is negative. This is unexpected.")
    return per


def round_batch(middle: int) -> int:
    if 2048 <= middle:
        middle = round(middle / 256) * 256
    if 512 <= middle < 2048:
        middle = round(middle / 64) * 64
    if 128 <= middle < 512:
        middle = round(middle / 16) * 16
    if 32 <= middle < 128:
        middle = round(middle / 4) * 4
    return middle


def find_next_batch_size(data, prefill_length, decode_length, available_memory) -> Optional[int]:
    """Middle ground between the largest batch that fit and the predicted 100 %-VRAM batch (or the smallest OOM)."""
    per = predict_memory_per_sample(data, prefill_length, decode_length)
    if per is None:
        return None
    ok = _rows(data, prefill_length, decode_length, oom=False)
    if not ok:
        raise ValueError("No successful measurements found.")
    b1, m1 = ok[-1]["batch_size"], ok[-1]["max_memory_allocated"]
    bad = _rows(data, prefill_length, decode_length, oom=True)
    b2 = bad[0]["batch_size"] if bad else b1 + int((available_memory - m1) / per)
    middle = round_batch((b1 + b2) // 2)
    if middle == b2 or middle <= b1:
        print("End of search")
        return None
    print(f"Found next batch size: {middle}")
    return middle


def main(argv=None):
    import pandas as pd
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-name", default="block_main_b4_1.2b")
    ap.add_argument("overrides", nargs="*", help="hydra-style ++key=value overrides")
    args = ap.parse_args(argv)
    cfg = apply_overrides(load_config(args.config_name), args.overrides)
    t0 = time.time()
    batch_sizes = cfg.get("benchmark_batch_sizes")
    P = cfg.get("benchmark_prefill_length", DEFAULT_PREFILL_LENGTH)
    D = cfg.get("benchmark_decode_length", DEFAULT_DECODE_LENGTH)
    # VRAM the batch search may fill.  The reference searches toward 100 % and stops below 2 GiB free (:445-468); on a
    # shared box the default here keeps 8 % head-room (an allocation failure inside a kernel launch or graph
    # instantiation is not a clean torch OOM).  ++benchmark_vram_fraction=1.0 restores the reference's rule;
    # ++benchmark_memory_limit_gib / ++benchmark_min_free_gib bound the search explicitly (tests).
    total_memory = torch.cuda.get_device_properties(DEVICE).total_memory / 1024 ** 3
    total_memory *= float(cfg.get("benchmark_vram_fraction", 0.92))
    if cfg.get("benchmark_memory_limit_gib") is not None:
        total_memory = float(cfg["benchmark_memory_limit_gib"])
    min_free = float(cfg.get("benchmark_min_free_gib", 2))
    output_path = os.path.join(cfg.get("save_dir") or SAVE_DIR, cfg.get("output_dir") or cfg["name"], "generation_time.csv")
    # ++profiling=true [++log_path=dir]: one trace file per measured batch size (reference :381-390, :460-463)
    log_dir = None
    if cfg.get("profiling", False):
        log_dir = cfg.get("log_path") or os.path.join(cfg.get("save_dir") or SAVE_DIR, "profiler")
        os.makedirs(log_dir, exist_ok=True)

    def trace_path(b):
        return None if log_dir is None else os.path.join(log_dir, f"{cfg['name']}_{b}_{P}_{D}.log")
    data: List[Dict] = []
    if os.path.exists(output_path):
        print("Reading existing measurement data")
        data = pd.read_csv(output_path, index_col=0).to_dict(orient="records")
    model = prepare_block_model(cfg, P, D)
    if batch_sizes is None:
        for b in (1, 2):
            if not measurement_exists(data, b, P, D):
                data.append(measure_generation(model, P, D, b, trace_path(b)))
        while total_memory - get_max_batch_size_row(data, P, D)["max_memory_allocated"] >= min_free:
            nxt = find_next_batch_size(data, P, D, total_memory)
            if nxt is None:
                break
            if predict_oom(data, nxt, P, D):
                raise AssertionError("Something is wrong with the measurements")
            data.append(measure_generation(model, P, D, nxt, trace_path(nxt)))
    else:
        for b in batch_sizes:
            if measurement_exists(data, b, P, D):
                print(f"Skipping batch size {b} as it is already measured")
            elif predict_oom(data, b, P, D):
                print(f"Skipping batch size {b} as it is predicted to OOM")
            else:
                data.append(measure_generation(model, P, D, b, trace_path(b)))
    df = pd.DataFrame(data).sort_values(["prefill_length", "decode_length", "batch_size"])
    os.makedirs(os.path.dirname(output_path), exist_ok=True)
    df.to_csv(output_path)
    print(f" Generation time data saved to {output_path} ".center(80, "-"))
    print(f"Total time: {time.time() - t0:.2f}s")
    return df


if __name__ == "__main__":
    main(sys.argv[1:])
