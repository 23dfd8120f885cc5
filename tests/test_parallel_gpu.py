"""Two REAL ranks (one process per GPU, RCCL) reproduce the single-process tokens — needs >= 2 visible MI355X, so it is
skipped on the 1-GPU test box and runs wherever the driver has a multi-GPU node."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rank_main(rank, world, port, ret):
    import numpy as np
    import torch.distributed as dist
    from block_transformer_amd import parallel
    from block_transformer_amd.config import named_dims
    from block_transformer_amd.model.utils import load_block_transformer_from_dims
    from oracle.cpu_ref import random_state_dict
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    parallel.init_distributed("nccl")
    assert dist.get_world_size() == world and dist.get_backend() == "nccl"
    torch.cuda.set_device(rank)
    dims = named_dims("micro")
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "micro.npz"))
    sd = random_state_dict(dims, seed=int(fx["seed"]), bias_std=float(fx["bias_std"]))
    model = load_block_transformer_from_dims(dims, "fp32")
    model.load_state_dict(sd, strict=True)
    model = model.to(f"cuda:{rank}")
    model.token_decoder.generation_config.eos_token_id = None
    prompt = torch.randint(1, dims.vocab, (7, 12), generator=torch.Generator().manual_seed(5))
    parallel.barrier()
    full = parallel.generate_sharded(lambda ids, **kw: model.generate(ids.to(f"cuda:{rank}"), **kw), prompt,
                                    pad_token_id=parallel.model_pad_token_id(model), max_length=28)
    ret[rank] = full.cpu()
    parallel.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_rccl_ranks_reproduce_single_rank_tokens():
    import torch.multiprocessing as mp
    from block_transformer_amd.config import named_dims
    from oracle.cpu_ref import oracle_from_dims, random_state_dict
    import numpy as np
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_main, args=(2, port, ret), nprocs=2, join=True)
    dims = named_dims("micro")
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "micro.npz"))
    sd = random_state_dict(dims, seed=int(fx["seed"]), bias_std=float(fx["bias_std"]))
    prompt = torch.randint(1, dims.vocab, (7, 12), generator=torch.Generator().manual_seed(5))
    want = oracle_from_dims(dims, sd).generate(prompt, max_length=28)
    assert torch.equal(ret[0], want) and torch.equal(ret[1], want)
