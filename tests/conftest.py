import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CONFIGS = ["micro", "micro_l8", "micro128", "micro64", "block_main_b4_5"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock comparison (sanity bound only; the ratio is printed, not gated)")


def load_golden(name):
    from block_transformer_amd.config import named_dims
    from oracle.cpu_ref import random_state_dict
    fx = dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))
    dims = named_dims(name)
    sd = random_state_dict(dims, seed=int(fx["seed"]), bias_std=float(fx["bias_std"]))
    return dims, sd, fx


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


@pytest.fixture
def dbg_lib():
    """Route the ops through libbt_hip_dbg.so (same kernels + bt_debug_* hooks) for one test; the product library
    libbt_hip.so is what every other test — and the product — runs."""
    from block_transformer_amd import _lib
    prev = _lib.use_debug_library(True)
    try:
        yield _lib.lib()
    finally:
        _lib.use_debug_library(prev)


def t(fx, key):
    return torch.from_numpy(np.asarray(fx[key]))


@pytest.fixture(scope="session")
def gpu_model():
    """Build the MI355X BlockTransformer for a golden config, load the fixture's weights, move to cuda."""
    cache = {}

    def get(name, dtype=torch.float32, **oracle_kw):
        key = (name, dtype)
        if key not in cache:
            from block_transformer_amd.model.utils import load_block_transformer_from_dims
            dims, sd, fx = load_golden(name)
            model = load_block_transformer_from_dims(dims, "fp32")
            model.load_state_dict(sd, strict=True)
            model = model.to(dtype).to("cuda")
            # the golden vectors were generated without EOS handling (benchmark protocol,
            # measure_generation_time.py:90); tests that exercise EOS set it explicitly
            model.token_decoder.generation_config.eos_token_id = None
            model.keep_logits = True    # tests look at the engine's last logits (default: greedy steps never write them)
            cache[key] = (model, dims, sd, fx)
        return cache[key]

    return get
