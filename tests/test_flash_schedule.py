"""Work distribution and ring / wait bookkeeping of the persistent flash-prefill kernel
(block_transformer_amd/csrc/attention.hip, flash_prefill_v3_kernel).

The kernel's workgroups walk a static list of (sequence-head group, query block) items.  A mapping bug would silently
skip or double-compute an output block, a ring-parity bug would let an LDS-DMA land on a tile that is still being read,
and a wrong `vmcnt` immediate would let a wave start an item before its operands are there.  This CPU model restates
the kernel's integer formulas (the expressions are checked to still be in the source, so the model cannot drift
silently) and checks, over a sweep of problem sizes:
  * every (group, query block) is processed exactly once, on the XCD that owns the group (group % 8), and the query
    blocks of one group run in the same round (so they share that XCD's L2);
  * a workgroup's items rotate through the causal block weights (static schedule, balanced to within the last round);
  * ring stages: the DMA issued in tile-iteration i never targets the stage read in iteration i, the output staging
    uses the last tile's stage, and the next item's prefetched first tile uses the other one;
  * the counted wait at an item's first tile covers exactly the prefetched tile's DMAs and the Q loads when the previous
    item issued a full block's stores, and falls back to vmcnt(0) otherwise."""
import os

import pytest

from conftest import ROOT

SRC = open(os.path.join(ROOT, "block_transformer_amd", "csrc", "attention.hip")).read()
FQ = 128        # query rows per item (FQ_ROWS)
FK = 64         # keys per tile (FK_TILE)


def test_model_formulas_are_the_kernels():
    for expr in ("const int sl = (S % nq == 0) ? ((k & 1) ? slot - slot % nq + (nq - 1 - slot % nq) : slot) : (slot + k) % S;",
                 "const int j = k * S + sl;",
                 "const int gi = j / nq, qi = j - gi * nq;",
                 "const int grp = gi * 8 + xcd;",
                 "qb = nq - 1 - qi;",
                 "const int lim = ((n_groups - xcd + 7) >> 3) * nq;",
                 "const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;",
                 "smem + ((gt + kt) & 1) * 2 * TILE_B",
                 "smem + ((gt + kt + 1) & 1) * 2 * TILE_B",
                 "smem + ((gt + n_tiles - 1) & 1) * 2 * TILE_B",
                 "constexpr int ST_FULL = FQ * SLOTS / (NW * 64);",
                 "pend_stores = ((qb + 1) * FQ <= q_len) ? ST_FULL : 0;",
                 "const int64_t per_xcd = ceil_div(n_groups, 8) * ceil_div(q_len, FQ_ROWS);",
                 "const int S = (int)std::min<int64_t>(std::max(1, 2 * flash_num_cus() / 8), per_xcd);"):
        assert expr in SRC, expr


def items_of_workgroup(p, S, n_groups, nq):
    """The (group, qb, round) items workgroup p processes, in order (item_of in the kernel)."""
    xcd, slot = p & 7, p >> 3
    lim = ((n_groups - xcd + 7) >> 3) * nq
    out = []
    k = 0
    while True:
        if S % nq == 0:
            sl = slot - slot % nq + (nq - 1 - slot % nq) if k & 1 else slot
        else:
            sl = (slot + k) % S
        j = k * S + sl
        if j >= lim:
            break
        gi, qi = divmod(j, nq)
        out.append((gi * 8 + xcd, nq - 1 - qi, k))
        k += 1
    return out


def grid_slots(n_groups, nq, cus=256):
    per_xcd = -(-n_groups // 8) * nq
    return min(max(1, 2 * cus // 8), per_xcd)


@pytest.mark.parametrize("n_groups", [1, 2, 4, 7, 8, 9, 16, 48, 100, 1024, 1030])
@pytest.mark.parametrize("q_len", [5, 64, 128, 130, 512, 700, 1024])
@pytest.mark.parametrize("cus", [256, 8, 304])
def test_every_block_once_on_its_xcd(n_groups, q_len, cus):
    nq = -(-q_len // FQ)
    S = grid_slots(n_groups, nq, cus)
    seen = {}
    rounds = {}
    for p in range(8 * S):
        for grp, qb, k in items_of_workgroup(p, S, n_groups, nq):
            assert 0 <= grp < n_groups and 0 <= qb < nq
            assert grp % 8 == p % 8                      # the workgroup's XCD owns the group
            assert (grp, qb) not in seen, (grp, qb)
            seen[(grp, qb)] = p
            rounds.setdefault(grp, set()).add(k)
    assert len(seen) == n_groups * nq
    if S % nq == 0:   # a round's S consecutive per-XCD items then hold whole groups: their blocks run together
        assert all(len(r) == 1 for r in rounds.values())
    else:             # otherwise a group straddles at most one more round boundary than its length forces
        assert all(len(r) <= -(-nq // S) + 1 for r in rounds.values())


def test_rotation_balances_causal_weights():
    """Prefill chunk of the benchmark: 64 sequences x 16 heads, 512 block positions (4 query blocks of 2/4/6/8 tiles)."""
    n_groups, nq = 1024, 4
    S = grid_slots(n_groups, nq)
    loads = []
    for p in range(8 * S):
        its = items_of_workgroup(p, S, n_groups, nq)
        assert len(its) == n_groups * nq // (8 * S)
        loads.append(sum(2 * qb + 2 for _, qb, _ in its))
        w = [2 * qb + 2 for _, qb, _ in its]
        assert all(w[i] + w[i + 1] == 10 for i in range(0, len(w) - 1, 2))   # heavy / light pairs: in step every two rounds
    assert max(loads) == min(loads)


@pytest.mark.parametrize("tiles", [[1], [2, 8, 1, 3], [1, 1, 1], [8, 6, 4, 2] * 3, [16, 1, 16]])
def test_ring_stages_and_prefetch(tiles):
    """Replay one workgroup's tile iterations: which stage is read, which is written, when."""
    gt = 0
    holds = {0: None, 1: None}          # stage -> (item, tile) whose K/V it currently holds
    holds[0] = (0, 0)                   # prologue: first item's tile 0 -> stage 0
    for it, n_tiles in enumerate(tiles):
        has_next = it + 1 < len(tiles)
        for kt in range(n_tiles):
            rd = (gt + kt) & 1
            assert holds[rd] == (it, kt), (it, kt, holds)      # the tile being computed is the one that landed there
            wr = (gt + kt + 1) & 1
            assert wr != rd
            # the barrier at the top of this iteration: everybody finished reading `wr` (the previous tile / the previous
            # item's output staging) -> it may be refilled now
            if kt + 1 < n_tiles:
                holds[wr] = (it, kt + 1)
            elif has_next:
                holds[wr] = (it + 1, 0)
        ost = (gt + n_tiles - 1) & 1
        assert ost == (gt + n_tiles - 1) & 1 and holds[ost] == (it, n_tiles - 1)   # staging overwrites the LAST tile only
        if has_next:
            assert holds[ost ^ 1] == (it + 1, 0)               # ... and never the prefetched first tile of the next item
        holds[ost] = "output"
        gt += n_tiles
        if has_next:
            assert holds[gt & 1] == (it + 1, 0)                # the next item finds its first tile at its parity


@pytest.mark.parametrize("D", [32, 64, 128])
def test_counted_wait_past_the_output_stores(D):
    """In-order vmcnt: ops issued by a wave between two items' first-tile waits, oldest first."""
    slots = D * 2 // 16
    st_full = FQ * slots // 256
    assert st_full in (2, 4, 8)                                # the three immediates the kernel has
    ni = 64 // (1024 // (D * 2)) // 4                          # DMA instructions per wave, operand and tile
    q_loads = 2 * (D // 32)
    for full_block in (True, False):
        queue = ["dma"] * (2 * ni) + ["q"] * q_loads           # next item's first tile, then its Q fragments
        n_st = st_full if full_block else 1                    # a block cut by q_len: some waves store fewer rows
        queue += ["st"] * n_st
        allowed = st_full if full_block else 0                 # pend_stores == ST_FULL ? vmcnt(ST_FULL) : vmcnt(0)
        done = queue[:max(0, len(queue) - allowed)]
        assert done.count("dma") == 2 * ni and done.count("q") == q_loads   # operands always complete
        if full_block:
            assert "st" not in done                            # and the store acknowledgements are not waited for
