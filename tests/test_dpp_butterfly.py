"""row16_sum_dpp (csrc/bt_common.h) against wave_sum<16>: the 16-lane butterfly of the local-window attention moved from four
ds_bpermute round trips (xor 8, 4, 2, 1) to four DPP adds (row_ror:8, row_ror:4, quad_perm [2,3,0,1], quad_perm [1,0,3,2]).
row_ror:4 does not deliver lane ^ 4 -- it delivers lane (i -+ 4) mod 16 -- so bit-identity of the SUM rests on an argument: after
the first stage lanes j and j ^ 8 hold the same value, hence the rotated partner holds the value lane ^ 4 holds, and fp32 addition
commutes.  Restated here in numpy fp32, stage by stage, for both rotation directions; the GPU tests (kernel vs fp64, fused vs unfused
bit-identity) run the real instruction sequence."""
import os
import re

import numpy as np

# DOCUMENTATION / PROOF test: the argument for bit-identity, on a numpy restatement -- not coverage of the HIP path (the
# GPU kernel tests are).  The DPP control words are parsed from bt_common.h so that a changed control code breaks this test instead
# of silently leaving the restatement behind: 0x128 = row_ror:8, 0x124 = row_ror:4, 0x4e = quad_perm [2,3,0,1], 0xb1 = quad_perm [1,0,3,2].
_HDR = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "block_transformer_amd", "csrc", "bt_common.h")).read()
_BODY = _HDR[_HDR.index("__device__ inline float row16_sum_dpp(float v)"):]
_BODY = _BODY[:_BODY.index("return v;")]
DPP_CTRL = [int(x, 16) for x in re.findall(r"__builtin_bit_cast\(int, v\), (0x[0-9a-f]+), 0xf, 0xf, false", _BODY)]
assert DPP_CTRL == [0x128, 0x124, 0x4E, 0xB1], DPP_CTRL


def _quad_perm(ctrl):
    """quad_perm control word -> source lane (within its quad) of each of the four lanes"""
    return [(ctrl >> (2 * i)) & 3 for i in range(4)]


assert _quad_perm(DPP_CTRL[2]) == [2, 3, 0, 1] and _quad_perm(DPP_CTRL[3]) == [1, 0, 3, 2]
assert [c - 0x120 for c in DPP_CTRL[:2]] == [8, 4]      # row_ror:N is encoded as 0x120 + N


def xor_butterfly(v):
    v = v.astype(np.float32).copy()
    lanes = np.arange(16)
    for o in (8, 4, 2, 1):
        v = (v + v[lanes ^ o]).astype(np.float32)
    return v


def dpp_butterfly(v, direction):
    v = v.astype(np.float32).copy()
    lanes = np.arange(16)
    for rot in (8, 4):                                   # row_ror:N -- lane i reads lane (i - N) mod 16 (or + N: same argument)
        v = (v + v[(lanes - direction * rot) % 16]).astype(np.float32)
    quad = lanes & ~3
    v = (v + v[quad + np.array([2, 3, 0, 1])[lanes & 3]]).astype(np.float32)   # quad_perm [2,3,0,1] == lane ^ 2
    v = (v + v[quad + np.array([1, 0, 3, 2])[lanes & 3]]).astype(np.float32)   # quad_perm [1,0,3,2] == lane ^ 1
    return v


def test_dpp_butterfly_is_bit_identical_to_the_xor_butterfly():
    rng = np.random.default_rng(1)
    for trial in range(2000):
        scale = 10.0 ** rng.integers(-6, 7)
        v = (rng.standard_normal(16) * scale).astype(np.float32)
        ref = xor_butterfly(v)
        assert np.all(ref == ref[0])                      # every lane ends with the row's sum
        for direction in (1, -1):
            got = dpp_butterfly(v, direction)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (trial, direction)


def test_the_argument_needs_the_first_stage():
    """Without the ^8 symmetry a rotation by 4 is NOT an xor-4 exchange: swapping the first two stages changes the pairing."""
    v = np.arange(16, dtype=np.float32) ** 3
    lanes = np.arange(16)
    a = (v + v[(lanes - 4) % 16]).astype(np.float32)     # rotation first
    b = (v + v[lanes ^ 4]).astype(np.float32)            # true xor-4 exchange
    assert not np.array_equal(a, b)
