"""Hazard model of the LDS half-tile rings of the 256x256 GEMM kernels (block_transformer_amd/csrc/gemm.hip).

The kernels overlap LDS-DMA (global_load_lds) with MFMA phases using counted `s_waitcnt vmcnt(N)`; a wrong count is a
silent race on hardware.  This CPU test re-derives, for every pipeline length, that
  RAW: every half-tile a phase reads was waited for by BOTH wave groups before the barrier that opens the reading epoch;
  WAR: a half-tile slot is refilled only after the barrier following its last read (for both staggered wave groups).
Constants (prefetch distances) are parsed from the source so the model cannot drift from the kernel."""
import os
import re

import pytest

from conftest import ROOT

SRC = open(os.path.join(ROOT, "block_transformer_amd", "csrc", "gemm.hip")).read()
PD = int(re.search(r"constexpr int PD = (\d+);", SRC).group(1))
PD2 = int(re.search(r"constexpr int PD2 = (\d+);", SRC).group(1))


def model(nk, phases_per_tile, pd):
    total = 4 * nk
    phases = phases_per_tile * nk
    if phases_per_tile == 4:
        need = lambda g: (g & ~3) + (1 if g % 4 == 0 else 2 if g % 4 == 1 else 3)
        reads_of = lambda g: {0: [4 * (g // 4), 4 * (g // 4) + 1], 1: [4 * (g // 4) + 2], 2: [4 * (g // 4) + 3], 3: []}[g % 4]
        last_read = lambda s: 4 * (s >> 2) + {0: 0, 1: 0, 2: 1, 3: 2}[s & 3]
        issued_thru = lambda g: min(g - 1 + pd, total - 1)          # loads issued through phase g-1
        loads_of_phase = lambda g: [g + pd]
        live = lambda g: g < total
    else:
        need = lambda g: 4 * (g >> 1) + (3 if g & 1 else 2)
        reads_of = lambda g: [4 * (g // 2) + j for j in range(3)] if g % 2 == 0 else [4 * (g // 2) + 3]
        last_read = lambda s: 2 * (s >> 2) + (1 if s & 3 == 3 else 0)
        issued_thru = lambda g: min(2 * (g - 1) + 1 + pd, total - 1)
        loads_of_phase = lambda g: [2 * g + pd, 2 * g + 1 + pd]
        live = lambda g: 4 * (g >> 1) < total
    issue_ep = {0: {}, 1: {}}
    waited = {0: {}, 1: {}}
    for grp in (0, 1):
        for s in range(min(pd, total)):
            issue_ep[grp][s] = -10
        for g in range(phases):
            for s in loads_of_phase(g):
                if s < total:
                    issue_ep[grp][s] = 2 * g + grp          # group 1 runs half a phase (one barrier) behind group 0

    def do_wait(grp, g, ep_end):
        if not live(g):
            return
        issued = issued_thru(g)
        allow = issued - need(g)
        assert allow >= 0, "needed load not even issued"
        assert issue_ep[grp][issued] <= ep_end
        keep = min(allow, 4)                                 # vmcnt(2*keep): the `keep` youngest half-tiles stay in flight
        for s in range(issued - keep + 1):
            waited[grp].setdefault(s, ep_end)

    for g in range(phases):
        do_wait(0, g, 2 * g - 1)                             # group 0 waits right before the barrier opening its phase g
    do_wait(1, 0, -1)
    for g in range(phases):
        do_wait(1, g + 1, 2 * g + 1)                         # group 1 waits before the barrier that opens group 0's phase g+1
    for g in range(phases):
        for grp, ep in ((0, 2 * g), (1, 2 * g + 1)):
            for s in reads_of(g):
                for w in (0, 1):
                    assert s in waited[w] and waited[w][s] < ep, f"RAW: nk={nk} phase {g} group {grp} load {s}"
    for s in range(8, total):                                # load s refills the slot of load s-8 (two K-tiles earlier)
        lr = last_read(s - 8)
        for w in (0, 1):
            assert 2 * lr + 1 < issue_ep[w][s], f"WAR: nk={nk} load {s} group {w}"


@pytest.mark.parametrize("nk", list(range(1, 34)) + [128])
def test_gemm_ring_schedules(nk):
    model(nk, 4, PD)
    model(nk, 2, PD2)


def test_model_detects_a_too_deep_prefetch():
    with pytest.raises(AssertionError, match="WAR"):
        model(8, 4, 9)
    with pytest.raises(AssertionError, match="WAR"):
        model(8, 2, 8)
