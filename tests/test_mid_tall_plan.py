"""K ranges of the 256 x 128 slab kernel (gemm_bf16_mfma256x128_slab_kernel, mid-batch stacks of 97 .. 512 rows; round 5).

DOCUMENTATION / PROOF test of the slicing, not of the HIP code: `tall_slices` (host) and the kernel's slice -> K-tile range are
restated here and checked for what the kernel and the reduce launches rely on -- the ranges of a pair partition its K-tiles, no range
is shorter than the 4 K-tiles the load stream's prologue assumes, a slice never straddles the two operand pairs, and the launch is
about one workgroup per CU.  The restated lines are asserted to be in the source so the two cannot drift; the kernel itself is
tested on the GPU (tests/test_model_gpu.py::test_mid_rows_*)."""
import os
import re

import pytest

from conftest import ROOT

SRC = open(os.path.join(ROOT, "block_transformer_amd", "csrc", "gemm.hip")).read()
WGS = int(re.search(r"BT_KNOB int g_mid_tall_wgs = (\d+);", SRC).group(1))
assert WGS == int(re.search(r"BT_KNOB int g_mid_tall_wgs_res = (\d+);", SRC).group(1))
for line in ("const int total = std::max(1, (nk2 > 0 ? g_mid_tall_wgs_res : g_mid_tall_wgs) / std::max(tiles, 1));",
             "*s1 = std::max(1, std::min(total, nk1 / 4));",
             "int a = (int)((int64_t)total * nk1 / (nk1 + nk2));",
             "a = std::max(1, std::min(a, std::max(1, nk1 / 4)));",
             "int b = std::max(1, std::min(total - a, std::max(1, nk2 / 4)));",
             "int kb = sidx * nk_all / scnt, ke = (sidx + 1) * nk_all / scnt;",
             "const int slice = id % S, tile = id / S;",
             "const int tm = tile / tiles_n, tn = tile - tm * tiles_n;",
             "const bool second = slice >= g.s1;"):
    assert line in SRC, line


def tall_slices(tiles, nk1, nk2):
    total = max(1, WGS // max(tiles, 1))
    if nk2 <= 0:
        return max(1, min(total, nk1 // 4)), 0
    a = total * nk1 // (nk1 + nk2)
    a = max(1, min(a, max(1, nk1 // 4)))
    b = max(1, min(total - a, max(1, nk2 // 4)))
    return a, b


def ranges(nk, s):
    return [(i * nk // s, (i + 1) * nk // s) for i in range(s)]


SHAPES = [  # (N, K1, K2): the dual QKV + h->4h launch is one problem of N1 + N2 columns as far as the slicing goes
    (6144 + 8192, 2048, 0), (2048, 8192, 2048), (2048, 2048, 8192),        # block_main_b4_1.2b
    (1536 + 2048, 512, 0), (512, 2048, 512),                               # the 512-wide test stack
    (3072 + 4096, 1024, 0), (1024, 4096, 1024), (768 * 3 + 3072, 768, 0), (768, 3072, 768),
    (128, 256, 0), (128, 256, 256), (50304, 2048, 0),
]


@pytest.mark.parametrize("N,K1,K2", SHAPES)
def test_k_ranges_partition_each_pair_and_respect_the_prologue(N, K1, K2):
    tiles, nk1, nk2 = N // 128, K1 // 64, K2 // 64
    s1, s2 = tall_slices(tiles, nk1, nk2)
    assert s1 >= 1 and (s2 >= 1) == (nk2 > 0)
    for nk, s in ((nk1, s1), (nk2, s2)):
        if not s:
            continue
        r = ranges(nk, s)
        assert r[0][0] == 0 and r[-1][1] == nk
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))                  # contiguous, disjoint
        assert min(e - b for b, e in r) >= min(4, nk), (nk, s, r)            # >= 4 K-tiles (host refuses K < 256)
        assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1     # balanced
    wgs = tiles * (s1 + s2)
    assert wgs <= max(WGS, tiles) + tiles                                    # about one workgroup per CU, never a second round by much


def test_two_row_tiles_halve_the_slices():
    """257 .. 512 rows: two row tiles per column tile -> the K split that fills the chip is half as deep (the QKV + h->4h launch needs
    none at all: 224 workgroups, no second slab)."""
    s, _ = tall_slices(2 * (6144 + 8192) // 128, 32, 0)
    assert s == 1
    s1, s2 = tall_slices(2 * 16, 128, 32)
    assert (s1, s2) == (6, 2) and 2 * 16 * (s1 + s2) == 256
    # slab bytes of the 1.2B layer at 512 rows (incl. the 64 x 64 kernels' two-slice worst case mid_slab_bytes also counts) fit the
    # stream-K scratch the engine's workspace carries above 256 rows -- asked of the library itself (host function, no GPU needed)
    from block_transformer_amd._lib import lib
    room = lib().bt_stream_k_scratch_bytes()
    assert max(2 * 512 * (6144 + 8192) * 4, (s1 + s2) * 512 * 2048 * 4) + 4096 <= room, room
    # ... and the engine's workspace of a bf16 stack above 256 rows does carry that scratch (fp32 stacks, which have no such
    # kernels, carry none: ADVICE r5) -- asked of bt_decoder_workspace_bytes itself
    import ctypes
    from block_transformer_amd import _lib
    dims = _lib.bt_decoder_dims(2048, 16, 128, 32, 8192, 12, 1e-5)
    acts = lambda rows, es: rows * (5 * 2048 + 8192) * es + 512       # noqa: E731  (activation buffers + alignment slack)
    assert lib().bt_decoder_workspace_bytes(ctypes.byref(dims), _lib.BT_BF16, 512) == acts(512, 2) + room
    assert lib().bt_decoder_workspace_bytes(ctypes.byref(dims), _lib.BT_BF16, 4096) == acts(4096, 2) + room
    assert lib().bt_decoder_workspace_bytes(ctypes.byref(dims), _lib.BT_F32, 512) == acts(512, 4)
    assert lib().bt_decoder_workspace_bytes(ctypes.byref(dims), _lib.BT_F32, 256) == acts(256, 4) + 16 * 256 * 8192 * 4


def test_the_1p2b_layer_fills_the_chip():
    s, _ = tall_slices((6144 + 8192) // 128, 32, 0)
    assert s == 2                                                            # 112 column tiles x 2 = 224 workgroups
    s1, s2 = tall_slices(16, 128, 32)                                        # mlp.dense_4h_to_h (K = 8192) + attention.dense (K = 2048)
    assert (s1, s2) == (12, 4) and 16 * (s1 + s2) == 256
    assert max(e - b for b, e in ranges(128, s1)) == 11 and max(e - b for b, e in ranges(32, s2)) == 8
