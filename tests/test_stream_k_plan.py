"""Stream-K item plan of gemm_bf16_mfma256sk_kernel (block_transformer_amd/csrc/gemm.hip, round 5), restated on the CPU.

DOCUMENTATION / PROOF test of the algorithm, not of the HIP code: `sk_plan` (host) and `SkWalk` (device: a workgroup's walk over
its run of items, chains in reverse tile order) are transliterated here and checked for the properties the kernel's hand-off relies
on.  The kernel itself is tested on the GPU (tests/test_kernels_gpu.py::test_stream_k_*: fp64, bit-identity of chained tiles,
determinism under load).  Constants are parsed from the source so the restatement cannot silently drift."""
import os
import re

import pytest

from conftest import ROOT

SRC = open(os.path.join(ROOT, "block_transformer_amd", "csrc", "gemm.hip")).read()
BM2 = int(re.search(r"constexpr int BM2 = (\d+), BN2 = (\d+);", SRC).group(1))
BK = int(re.search(r"constexpr int BM = \d+, BN = \d+, BK = (\d+);", SRC).group(1))
GROUP_M2 = int(re.search(r"constexpr int GROUP_M2 = (\d+);", SRC).group(1))
assert "for (int C = 1; C <= nkt / 4; ++C)" in SRC and "if (nk < 4 || nk1 % nk != 0) continue;" in SRC   # what sk_plan below restates
assert "tile = (hi - 1) / C;" in SRC and "cs = max(lo - tile * C, 0);" in SRC                                # ... and SkWalk


def sk_plan(M, N, K1, K2, G):
    if G < 8 or G % 8 or M % BM2 or N % BM2 or K1 % BK or K2 % BK or K1 <= 0:
        return None
    T = (M // BM2) * (N // BM2)
    nk1, nkt = K1 // BK, (K1 + K2) // BK
    if T < G // 2:
        return None
    for C in range(1, nkt // 4 + 1):
        if nkt % C or (T * C) % G:
            continue
        nk = nkt // C
        if nk < 4 or nk1 % nk:
            continue
        q = T * C // G
        if all(((t * C + C - 1) // q - (t * C) // q) <= 1 for t in range(T)):
            return dict(nk=nk, C=C, q=q, T=T)
    return None


def walk(lo, q, C):
    """SkWalk: the items (tile, chunk, chain_first, chain_last, publish, consume) of the run [lo, lo + q) in the kernel's order."""
    hi = lo + q
    tile = (hi - 1) // C
    cs, ce = max(lo - tile * C, 0), hi - tile * C
    c = cs
    out = []
    for _ in range(q):
        first, last = c == cs, c == ce - 1
        publish = last and ce != C
        consume = last and not publish and cs != 0
        out.append((tile, c, first, last, publish, consume))
        c += 1
        if c == ce:
            tile -= 1
            ce = C
            cs = max(lo - tile * C, 0)
            c = cs
    return out


def tile_coords(tile, tiles_m, tiles_n):
    group_sz = GROUP_M2 * tiles_n
    g = tile // group_sz
    first_m = g * GROUP_M2
    gm = min(tiles_m - first_m, GROUP_M2)
    in_g = tile - g * group_sz
    return (first_m + in_g % gm), in_g // gm


SHAPES = [
    (4096, 6144, 2048, 0, dict(nk=16, C=2, q=3)),        # QKV at the benchmark batch: 384 tiles = 1.5 rounds
    (4096, 2048, 8192, 2048, dict(nk=16, C=10, q=5)),    # fused residual projections, decode step
    (8192, 2048, 8192, 2048, dict(nk=32, C=5, q=5)),     # token step 0 (B * E rows)
    (32768, 2048, 8192, 2048, dict(nk=32, C=5, q=20)),   # a prefill chunk
    (6144, 6144, 2048, 0, dict(nk=8, C=4, q=9)),
    (6144, 2048, 8192, 2048, dict(nk=8, C=20, q=15)),
    (3072, 6144, 2048, 0, dict(nk=4, C=8, q=9)),
    (4096, 2048, 2048, 512, dict(nk=4, C=10, q=5)),
    (2048, 4096, 1024, 1024, dict(nk=16, C=2, q=1)),
    (2304, 8192, 2048, 0, dict(nk=4, C=8, q=9)),
]


@pytest.mark.parametrize("M,N,K1,K2,want", SHAPES)
def test_plan_and_walk_invariants(M, N, K1, K2, want):
    G = 256
    p = sk_plan(M, N, K1, K2, G)
    assert p is not None and {k: p[k] for k in want} == want, p
    C, q, T, nk = p["C"], p["q"], p["T"], p["nk"]
    nk1 = K1 // BK
    seen = {}
    publishers, consumers = {}, {}
    for vw in range(G):
        items = walk(vw * q, q, C)
        assert len(items) == q
        # (1) chain structure: accumulators carry from an item to the next only inside one tile, chunks ascending
        for a, b in zip(items, items[1:]):
            if not a[3]:
                assert b[0] == a[0] and b[1] == a[1] + 1 and not b[2]
            else:
                assert b[2] and b[0] == a[0] - 1, "chains are walked in reverse tile order"
        assert items[0][2] and items[-1][3]
        # (2) a publish chain can only be the FIRST chain of a workgroup, a consume chain only the LAST
        kinds = [("P" if it[4] else "C" if it[5] else "E") for it in items if it[3]]
        assert "P" not in kinds[1:] and "C" not in kinds[:-1], kinds
        for tile, c, first, last, publish, consume in items:
            assert (tile, c) not in seen, "a unit is computed twice"
            seen[(tile, c)] = vw
            k0 = c * nk
            assert (k0 < nk1) == (k0 + nk <= nk1), "an item straddles the two operand pairs"
            if publish:
                publishers[tile] = vw
            if consume:
                consumers[tile] = vw
            if first and c > 0:
                assert any(it[5] for it in items if it[0] == tile), "a chain that starts inside its tile must consume the head's partial"
    assert len(seen) == T * C, "not every (tile, chunk) unit is computed"
    # (3) the hand-off is pairwise and goes to the NEXT virtual workgroup (slab / flag index vw + 1 in the kernel)
    assert set(publishers) == set(consumers)
    for tile, vw in publishers.items():
        assert consumers[tile] == vw + 1
    assert len(set(consumers.values())) == len(consumers), "a workgroup consumes at most one tile (one slab per workgroup)"
    # (4) tile -> coordinates is a bijection onto the tile grid
    tm_n, tn_n = M // BM2, N // BM2
    assert len({tile_coords(t, tm_n, tn_n) for t in range(T)}) == T
    assert all(0 <= a < tm_n and 0 <= b < tn_n for a, b in (tile_coords(t, tm_n, tn_n) for t in range(T)))


def test_hand_off_is_off_the_critical_path_for_one_and_a_half_rounds():
    """M = 4096 QKV: the publisher writes its partial after ONE of its three items, the consumer reads it after its third."""
    p = sk_plan(4096, 6144, 2048, 0, 256)
    for vw in range(0, 256, 2):
        pub, con = walk(vw * p["q"], p["q"], p["C"]), walk((vw + 1) * p["q"], p["q"], p["C"])
        assert pub[0][4] and [it[4] for it in pub[1:]] == [False, False]
        assert con[-1][5] and pub[0][0] == con[-1][0]


def test_plans_that_must_be_refused():
    assert sk_plan(4096, 6144, 2048, 0, 250) is None          # G % 8
    assert sk_plan(4000, 6144, 2048, 0, 256) is None          # ragged rows
    assert sk_plan(1024, 2048, 8192, 2048, 256) is None       # fewer tiles than half the chip
    assert sk_plan(4352, 2048, 8192, 2048, 256) is None       # 136 tiles: no chunking divides evenly over 256 workgroups
