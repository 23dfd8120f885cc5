"""N > 1 path on CPU: world_size-2 gloo processes exercise the request sharding, the barrier, the max/sum reductions
and the id gather — with a stand-in generate function (the HIP engine needs a GPU; shard-equivalence of the real
engine is covered by tests/test_model_gpu.py::test_generate_shard_equivalence)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from block_transformer_amd import parallel


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 4096):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_bounds(4, 2, 2)


def _fake_generate(ids, max_length):
    """Deterministic per-request 'generation': depends only on the request's own prompt (like the real engine)."""
    B, P = ids.shape
    out = torch.zeros(B, max_length, dtype=torch.int64)
    out[:, :P] = ids
    for tcol in range(P, max_length):
        out[:, tcol] = (out[:, :tcol].sum(1) * 31 + tcol) % 977
    return out


def _worker(rank, world, port, n_req, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed("gloo")
    assert dist.get_world_size() == world
    ids = torch.randint(1, 500, (n_req, 6), generator=torch.Generator().manual_seed(0))
    parallel.barrier()
    full = parallel.generate_sharded(_fake_generate, ids, max_length=11)
    mx = parallel.max_over_ranks(1.0 + rank)
    sm = parallel.sum_over_ranks(10.0 * (rank + 1))
    shard = parallel.shard_requests(ids, rank, world)
    ret[rank] = (full, mx, sm, shard.shape[0])
    parallel.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_req", [5, 8])
def test_two_process_gloo_sharding(n_req):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, n_req, ret), nprocs=2, join=True)
    ids = torch.randint(1, 500, (n_req, 6), generator=torch.Generator().manual_seed(0))
    want = _fake_generate(ids, 11)
    for r in (0, 1):
        full, mx, sm, nshard = ret[r]
        assert torch.equal(full, want), "sharded output must equal the single-process output, in request order"
        assert mx == 2.0 and sm == 30.0
    assert ret[0][3] + ret[1][3] == n_req
