"""Data-parallel request sharding for the decode engine: one process per GPU, independent request shards.

The decode path shards naturally: sequences share no state, weights are replicated read-only, so the
steady state has NO data-path collective.  ``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm, "gloo" on
CPU for tests) is used only for: the launch/timing barrier, the max-over-ranks elapsed time, the sum of generated
tokens, and (optionally) gathering the generated ids on rank 0."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0)))


def parse_cpulist(text: str) -> List[int]:
    """Linux cpulist syntax ("0-15,32-47", "3", "") -> sorted CPU ids."""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.update(range(int(lo), int(hi) + 1))
        else:
            cpus.add(int(part))
    return sorted(cpus)


def _is_amdgpu(dev_dir: str) -> bool:
    """True when a /sys/class/drm/card*/device directory is an AMD GPU function: bound to the amdgpu driver, or PCI vendor 0x1002."""
    try:
        if os.path.basename(os.path.realpath(os.path.join(dev_dir, "driver"))) == "amdgpu":
            return True
    except OSError:
        pass
    try:
        return int(open(os.path.join(dev_dir, "vendor")).read().strip(), 16) == 0x1002
    except (OSError, ValueError):
        return False


def gpu_numa_cpus(local_rank: int, sysfs_root: str = "/sys") -> Tuple[Optional[int], List[int]]:
    """(NUMA node, CPU ids local to it) of the GPU ``local_rank`` drives, from sysfs: the PCI function of the torch device
    (``pci_domain_id:pci_bus_id:pci_device_id.0``) when torch exposes it, else the local_rank-th ``/sys/class/drm/card*`` that
    is an amdgpu device.  (None, []) when the box does not say (single-node container without NUMA information)."""
    dev_dirs = []
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
        dev_dirs.append(os.path.join(sysfs_root, "bus", "pci", "devices", bdf))
    except Exception:
        pass
    drm = os.path.join(sysfs_root, "class", "drm")
    try:
        cards = sorted((c for c in os.listdir(drm) if c.startswith("card") and c[4:].isdigit()), key=lambda c: int(c[4:]))
        cards = [os.path.join(drm, c, "device") for c in cards if os.path.exists(os.path.join(drm, c, "device", "numa_node"))]
        cards = [d for d in cards if _is_amdgpu(d)]     # not a BMC / ASPEED VGA function that also has a numa_node file
        if local_rank < len(cards):
            dev_dirs.append(cards[local_rank])
    except OSError:
        pass
    for d in dev_dirs:
        try:
            node = int(open(os.path.join(d, "numa_node")).read().strip())
        except (OSError, ValueError):
            continue
        if node < 0:
            return None, []
        for src in (os.path.join(d, "local_cpulist"), os.path.join(sysfs_root, "devices", "system", "node", f"node{node}", "cpulist")):
            try:
                cpus = parse_cpulist(open(src).read())
            except (OSError, ValueError):
                continue
            if cpus:
                return node, cpus
        return node, []
    return None, []


def bind_to_gpu_numa_node(local_rank: int, local_world: int, sysfs_root: str = "/sys") -> Optional[dict]:
    """Pin this rank's host threads to the CPUs of its GPU's NUMA node (small batches are host-launch bound: 52 launches per
    block step at B = 1, and a rank scheduled on the far socket pays cross-socket latency on every one).  Ranks sharing a node
    split its CPUs evenly by local rank.  ``local_world`` = the number of ranks on THIS node (LOCAL_WORLD_SIZE; the peer scan walks
    local ranks only -- a multi-node job's global world size would index GPUs this node does not have).  The chosen binding is
    logged once to stderr so a bad pin is visible.  sched_setaffinity(0) covers the calling thread and threads created later: call
    it before the process group (and its helper threads) comes up, as init_distributed does.  No-op (returns None) when sysfs has
    no NUMA information, the intersection with the current affinity mask is empty, or BT_NO_NUMA_BIND=1."""
    if os.environ.get("BT_NO_NUMA_BIND") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    node, cpus = gpu_numa_cpus(local_rank, sysfs_root)
    if node is None or not cpus:
        return None
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    peers = [r for r in range(max(1, local_world)) if gpu_numa_cpus(r, sysfs_root)[0] == node] or [local_rank]
    if local_rank in peers and len(peers) > 1 and len(allowed) >= len(peers):
        per = len(allowed) // len(peers)
        i = peers.index(local_rank)
        allowed = allowed[i * per:(i + 1) * per]
    os.sched_setaffinity(0, allowed)
    import sys
    print(f"[bt parallel] local rank {local_rank}: bound to NUMA node {node}, {len(allowed)} CPU(s) "
          f"{allowed[0]}..{allowed[-1]} ({len(peers)} local rank(s) share the node)", file=sys.stderr, flush=True)
    return {"numa_node": node, "cpus": allowed}


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, world, local = env_world()
    if world > 1:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", 0)) or (torch.cuda.device_count() if torch.cuda.is_available() else world)
        bind_to_gpu_numa_node(local, min(world, local_world))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced shard [lo, hi) of ``n_items`` requests for ``rank`` (first n % world ranks get one more)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_requests(input_ids: torch.Tensor, rank: int, world: int, *others: Optional[torch.Tensor]):
    lo, hi = shard_bounds(input_ids.shape[0], rank, world)
    out = [input_ids[lo:hi]] + [None if o is None else o[lo:hi] for o in others]
    return out[0] if not others else tuple(out)


def _device_for_collectives() -> torch.device:
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def barrier() -> None:
    """Launch barrier: one tiny all-reduce (RCCL over xGMI on the GPU box)."""
    if dist.is_initialized():
        t = torch.zeros(1, dtype=torch.int32, device=_device_for_collectives())
        dist.all_reduce(t)
        if t.is_cuda:
            torch.cuda.synchronize()


def max_over_ranks(value: float) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_device_for_collectives())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_device_for_collectives())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(value: float) -> List[float]:
    """Every rank's value, in rank order, on every rank ([value] without a process group)."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=_device_for_collectives())
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def gather_ids(local_ids: torch.Tensor, n_total: int, pad_token_id: int) -> Optional[torch.Tensor]:
    """Concatenate every rank's generated ids (shard order == request order); returned on every rank.

    Shards may come back with different widths: with EOS handling enabled each shard's generate() stops as soon as all
    of ITS sequences are finished, and an empty shard has no width of its own.  The common width is agreed with one
    MAX all-reduce and narrower shards are right-padded with ``pad_token_id`` before the all-gather."""
    if not dist.is_initialized():
        return local_ids
    world = dist.get_world_size()
    dev = _device_for_collectives()
    w = torch.tensor([local_ids.shape[1] if local_ids.dim() == 2 and local_ids.shape[0] > 0 else 0], dtype=torch.int64,
                     device=dev)
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    width = int(w.item())
    per = -(-n_total // world)
    buf = torch.full((per, width), int(pad_token_id), dtype=torch.int64, device=dev)
    if local_ids.shape[0] > 0:
        buf[: local_ids.shape[0], : local_ids.shape[1]] = local_ids.to(dev)
    parts: List[torch.Tensor] = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        out.append(parts[r][: hi - lo])
    return torch.cat(out, dim=0)


def model_pad_token_id(model) -> int:
    """The id a single-process ``BlockTransformer.generate`` writes behind a finished sequence: whole blocks of
    ``token_decoder.config.pad_token_id`` (reference model/block_transformer.py:384) -- what a narrower shard must be
    right-padded with so that the sharded result equals the unsharded one."""
    return int(model.token_decoder.config.pad_token_id)


def generate_sharded(generate_fn, input_ids: torch.Tensor, *, pad_token_id: int, **kw) -> torch.Tensor:
    """Run ``generate_fn`` on this rank's request shard and gather the full result (right-padded to the widest shard).

    ``pad_token_id`` is required: shards stop independently under EOS handling, and the tail of a narrower
    shard must carry the id the unsharded generate() would have written there -- ``model_pad_token_id(model)``."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    lo, hi = shard_bounds(input_ids.shape[0], rank, world)
    local = generate_fn(input_ids[lo:hi], **kw) if hi > lo else input_ids.new_zeros(0, 0)
    return gather_ids(local, input_ids.shape[0], pad_token_id)


def launch_self_under_torchrun(n_procs: int, script: str, argv: List[str]) -> None:
    """Re-exec ``script argv`` as ``n_procs`` ranks of one node under ``torch.distributed.run`` when this process is not
    already one of them.  ``--standalone`` lets the launcher itself pick (and hold) a free rendezvous port on 127.0.0.1 --
    no bind-then-close probe that a concurrent launch could race.  Does not return on success (os.execv)."""
    import sys
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or n_procs <= 1:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n_procs}", script] + list(argv)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)
