"""The in-place ascending gather behind bt_compact_enqueue (csrc/misc.hip: compact_kv_kernel), restated on the CPU as the per-thread
schedule the kernel executes -- a thread owns ONE 16-byte offset of every sequence slot and walks the live list upward in batches of
COMPACT_U: all loads of a batch first (slots live[i .. i+U)), then its stores (slots i .. i+U).  The claim the kernel rests on: no
load ever reads a slot an earlier store of the same walk has overwritten, whatever the ascending live list, so no temporary is
needed.  Checked here exhaustively for small sizes and on random lists; the GPU test (test_compact_enqueue_c_abi) checks the kernel."""
import itertools
import os
import re

import numpy as np
import pytest

# DOCUMENTATION / PROOF test: it checks the ARGUMENT the kernel rests on, on a numpy restatement; it never runs product
# code and is not coverage of the HIP path (tests/test_kernels_gpu.py::test_compact_enqueue_c_abi and the compaction tests of
# tests/test_model_gpu.py are).  The batch size is parsed from the kernel source so that the restatement follows a COMPACT_U change.
_SRC = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "block_transformer_amd", "csrc", "misc.hip")).read()
COMPACT_U = int(re.search(r"constexpr int COMPACT_U = (\d+);", _SRC).group(1))
assert "for (; i < n_live; i += COMPACT_U)" in _SRC   # loads of a batch first, then its stores: the schedule restated below


def walk(slots: np.ndarray, live, u: int = COMPACT_U):
    """One thread's walk over its offset: `slots[s]` = the 16 bytes it owns in sequence slot s.  Returns the slots after the walk
    and the set of slots that were read AFTER having been overwritten (must be empty)."""
    slots = slots.copy()
    written, stale_reads = set(), []
    n = len(live)
    i = 0
    while i < n and live[i] == i:      # rows already in place
        i += 1
    while i < n:
        batch = range(i, min(i + u, n))
        regs = {}
        for j in batch:                # loads first ...
            if live[j] in written and live[j] != j:
                stale_reads.append((j, live[j]))
            regs[j] = slots[live[j]]
        for j in batch:                # ... then the stores
            slots[j] = regs[j]
            written.add(j)
        i += u
    return slots, stale_reads


@pytest.mark.parametrize("B", [1, 2, 5, 9])
def test_every_ascending_live_list_small(B):
    data = np.arange(100, 100 + B)
    for n in range(0, B + 1):
        for live in itertools.combinations(range(B), n):
            for u in (1, 2, COMPACT_U):
                out, stale = walk(data, list(live), u)
                assert not stale, (live, u, stale)
                assert np.array_equal(out[:n], data[list(live)]), (live, u)
                assert np.array_equal(out[n:], data[n:])      # slots >= n_live are never written


def test_random_live_lists_large():
    rng = np.random.default_rng(0)
    for _ in range(300):
        B = int(rng.integers(1, 600))
        n = int(rng.integers(0, B + 1))
        live = np.sort(rng.choice(B, size=n, replace=False)).tolist()
        data = rng.integers(0, 1 << 30, size=B)
        out, stale = walk(data, live)
        assert not stale
        assert np.array_equal(out[:n], data[live])


def test_a_descending_or_repeated_list_would_break_it():
    """The precondition matters (bt_compact_enqueue documents it): with a non-ascending list the walk does read overwritten slots."""
    data = np.arange(10)
    _, stale = walk(data, [3, 0, 1], u=1)      # slot 0 is overwritten at step 0 and read at step 1
    assert stale
