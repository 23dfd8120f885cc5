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


# ---- persistent kernel (gemm_bf16_mfma256pp_kernel) -----------------------------------------------------------------------
# Literal program-order model: every wave's VMEM queue retires IN ORDER (gfx9 vmcnt); `s_waitcnt vmcnt(N)` completes all but
# the N youngest ops.  The kernel's counted waits add PP_STORES after a FULL tile's epilogue and nothing after an edge tile.
PP_STORES = int(re.search(r"constexpr int PP_STORES = (\d+);", SRC).group(1))


def _pp_wait_imm(g, total, stores_younger, pd):
    """transliteration of wait_for_phase_pp: the vmcnt immediate, or None when no wait is issued"""
    t = g >> 1
    if 4 * t >= total:
        return None
    need = 4 * t + (3 if g & 1 else 2)
    issued = min(2 * (g - 1) + 1 + pd, total - 1)
    allow = issued - need
    assert allow >= 0
    return 2 * min(allow, 4) + (PP_STORES if stores_younger else 0)


def _pp_stream(grp, wave0, nk, nt, full, stores, pd, has_bias):
    """program-order event list of one wave: ('vm', tag) VMEM op, ('wait', N), ('bar',), ('read', tag)"""
    total = 4 * nk * nt
    ev = []

    def dma(s):
        if s < total:
            if has_bias and wave0 and s % (4 * nk) == 0:
                ev.append(("vm", ("bias", s // (4 * nk))))   # issue_bias() right before the new tile's first half-tile
            ev.append(("vm", ("ld", s)))
            ev.append(("vm", ("ld", s)))                     # two DMA instructions per half-tile

    for s in range(pd):
        dma(s)
    if grp == 1:
        n = _pp_wait_imm(0, total, False, pd)
        if n is not None:
            ev.append(("wait", n))
        ev.append(("bar",))
    g = 0
    prev_full = False
    for j in range(nt):
        for t in range(nk):
            for p in (0, 1):
                if grp == 0:
                    n = _pp_wait_imm(g, total, prev_full and t == 0, pd)
                    if n is not None:
                        ev.append(("wait", n))
                ev.append(("bar",))
                if p == 0 and t == 0 and has_bias:
                    ev.append(("read", ("bias", j)))
                x = g >> 1
                for s in ([4 * x, 4 * x + 1, 4 * x + 2] if p == 0 else [4 * x + 3]):
                    ev.append(("read", ("ld", s)))
                dma(2 * g + pd)
                dma(2 * g + 1 + pd)
                if grp == 1:
                    n = _pp_wait_imm(g + 1, total, prev_full and t == 0 and p == 0, pd)
                    if n is not None:
                        ev.append(("wait", n))
                ev.append(("bar",))
                g += 1
        for _ in range(stores[j]):
            ev.append(("vm", ("st", j)))
        prev_full = full[j]
    if grp == 0:
        ev.append(("bar",))
    return ev


def _pp_analyse(ev):
    """-> (issue_bar[tag], done_bar[tag], reads[(tag, bar)]) with bar = barriers executed before the event"""
    bars, queue, done_upto = 0, [], 0
    issue_bar, done_bar, reads = {}, {}, []
    for e in ev:
        if e[0] == "bar":
            bars += 1
        elif e[0] == "vm":
            queue.append(e[1])
            issue_bar.setdefault(e[1], bars)
            issue_bar[("last", e[1])] = bars
        elif e[0] == "wait":
            upto = max(done_upto, len(queue) - e[1])
            for tag in queue[done_upto:upto]:
                done_bar[tag] = bars            # (a half-tile counts as landed when its SECOND instruction retired)
            done_upto = upto
        else:
            reads.append((e[1], bars))
    return issue_bar, done_bar, reads, bars


def pp_model(nk, nt, full, stores, pd=None, has_bias=True, stream=None):
    pd = PD2 if pd is None else pd
    stream = stream or _pp_stream
    waves = {(0, True): None, (0, False): None, (1, False): None}   # wave 0 (issues the bias DMA), other group-0 waves, group 1
    for (grp, w0) in waves:
        waves[(grp, w0)] = _pp_analyse(stream(grp, w0, nk, nt, full, stores, pd, has_bias))
    nbars = {v[3] for v in waves.values()}
    assert len(nbars) == 1, "barrier counts differ between wave groups"
    last_read = {}
    for key, (issue, done, reads, _) in waves.items():
        for tag, b in reads:
            last_read[tag] = max(last_read.get(tag, -1), b)
            for okey, (oissue, odone, _, _) in waves.items():
                if tag[0] == "bias" and not okey[1]:
                    continue                     # only wave 0 issued (and must have waited for) the bias DMA
                assert tag in odone and odone[tag] <= b - 1, f"RAW nk={nk} nt={nt}: {tag} read by {key} not waited by {okey}"
    for key, (issue, done, reads, _) in waves.items():
        for tag, b in list(issue.items()):
            if tag[0] == "ld" and tag[1] >= 8:
                prev = ("ld", tag[1] - 8)
            elif tag[0] == "bias" and tag[1] >= 2:
                prev = ("bias", tag[1] - 2)
            else:
                continue
            assert b >= last_read[prev] + 1, f"WAR nk={nk} nt={nt}: {tag} issued by {key} before the last read of {prev}"


@pytest.mark.parametrize("nk", [4, 5, 8, 32])
@pytest.mark.parametrize("nt", [1, 2, 3, 5])
def test_persistent_gemm_schedule(nk, nt):
    pp_model(nk, nt, [True] * nt, [PP_STORES] * nt)                 # interior tiles
    pp_model(nk, nt, [True] * nt, [2 * PP_STORES] * nt)             # head_dim 32 QKV scatter: more stores than assumed
    pp_model(nk, nt, [False] * nt, [0] * nt)                        # edge tiles whose rows are all out of range for a wave
    pp_model(nk, nt, [False] * nt, [PP_STORES] * nt)                # edge tile, but this wave's rows are all valid
    pp_model(nk, nt, [j % 2 == 0 for j in range(nt)], [PP_STORES if j % 2 == 0 else 3 for j in range(nt)], has_bias=False)


def test_persistent_model_detects_miscounted_stores():
    # claiming PP_STORES younger stores when the wave issued fewer must be caught (this is the one unsafe direction)
    with pytest.raises(AssertionError, match="RAW"):
        pp_model(8, 3, [True] * 3, [PP_STORES - 8] * 3)
    with pytest.raises(AssertionError, match="WAR"):
        pp_model(8, 3, [True] * 3, [PP_STORES] * 3, pd=8)



# ---- second-generation K-loop (gemm_bf16_mfma256pp2_kernel): constant vmcnt immediates outside the stream tail -----------------
PP2_WAIT = [int(v) for v in re.search(r"constexpr int PP2_WAIT_P0 = (\d+), PP2_WAIT_P1 = (\d+);", SRC).groups()]


@pytest.mark.parametrize("nk", [4, 5, 8, 32, 128])
@pytest.mark.parametrize("nt", [1, 2, 3, 12])
def test_pp2_constant_waits_equal_the_generic_function(nk, nt):
    """The v2 kernel hard-codes vmcnt(PP2_WAIT_P0 / P1) (+ PP_STORES in K-tile 0 behind a full tile) for every phase except the
    last two K-tiles of the workgroup's last tile, where it calls the generic function.  For exactly those phases the
    generic function (the one the hazard model above verifies) must evaluate to the same immediates -- for group 0's wait
    for its own phase and for group 1's wait for the next phase."""
    total = 4 * nk * nt
    g = 0
    for j in range(nt):
        t_gen = nk - 2 if j == nt - 1 else nk
        for t in range(nk):
            for p in (0, 1):
                if t < t_gen:
                    for prev_full in (False, True):
                        sy0 = prev_full and t == 0
                        assert _pp_wait_imm(g, total, sy0, PD2) == PP2_WAIT[p] + (PP_STORES if sy0 else 0), (nk, nt, j, t, p)
                        sy1 = prev_full and t == 0 and p == 0
                        assert _pp_wait_imm(g + 1, total, sy1, PD2) == PP2_WAIT[1 - p] + (PP_STORES if sy1 else 0), (nk, nt, j, t, p)
                g += 1


def _pp2_stream(grp, wave0, nk, nt, full, stores, pd, has_bias, bad=None, dmac=False, drain=()):
    """Program-order event list of one wave of gemm_bf16_mfma256pp2_kernel: constant waits in K-tiles >= 2, store-count waits in
    K-tiles 0 / 1 behind a full tile, W1 / A1 of the next tile's K-tile 1 issued at the tile boundary behind one extra barrier and
    AHEAD of the tile's stores; generic waits in the last two K-tiles of the last tile.  `bad` injects a deliberate error."""
    assert pd == PD2
    total = 4 * nk * nt
    ev = []
    cursor = [0]                                  # next half-tile load to issue (consumption order)

    def dma2(n=2):                                # the next n half-tiles
        for _ in range(n):
            sidx = cursor[0]
            cursor[0] += 1
            if sidx < total:
                if has_bias and wave0 and sidx % (4 * nk) == 0:
                    ev.append(("vm", ("bias", sidx // (4 * nk))))
                ev.append(("vm", ("ld", sidx)))
                ev.append(("vm", ("ld", sidx)))

    for _ in range(3):
        dma2()                                    # prologue: PD2 = 6 half-tiles
    if grp == 1:
        n = _pp_wait_imm(0, total, False, pd)
        if n is not None:
            ev.append(("wait", n))
        ev.append(("bar",))
    g = 0
    prev_full = False
    pre_issued = False
    dmac_all = dmac
    fast_kt = set()
    if dmac == "fast":                            # the kernel's experiment: only in the fast K-tiles of full tiles
        fast_kt = {(j, t) for j, t, kind in _pp2_ktile_walk(nk, nt, full)[0] if kind == "fast"}
    for j in range(nt):
        t_gen = nk - 2 if j == nt - 1 else nk
        for t in range(nk):
            pm = 0 if t < 2 else (1 if t < t_gen else 2)
            dmac = (dmac_all is True) or ((j, t) in fast_kt)
            for p in (0, 1):
                if grp == 0:
                    if pm == 2:
                        n = _pp_wait_imm(g, total, False, pd)
                    else:
                        n = PP2_WAIT[p] + (PP_STORES if (pm == 0 and prev_full) else 0)
                    if n is not None:
                        ev.append(("wait", n))
                ev.append(("bar",))
                if p == 0 and t == 0 and has_bias:
                    ev.append(("read", ("bias", j)))
                x = g >> 1
                for sidx in ([4 * x, 4 * x + 1, 4 * x + 2] if p == 0 else [4 * x + 3]):
                    ev.append(("read", ("ld", sidx)))
                late = False
                if p == 0 and t == 0 and pre_issued:
                    pre_issued = False
                elif dmac:
                    dma2(1)
                    late = True
                else:
                    dma2()
                if grp == 1:
                    if pm == 2:
                        n = _pp_wait_imm(g + 1, total, False, pd)
                        if n is not None and dmac:
                            n = max(n - 2, 0)
                    else:
                        sy = pm == 0 and prev_full and not (t == 1 and p == 1)
                        if bad == "late_store_wait":
                            sy = pm == 0 and prev_full
                        n = PP2_WAIT[1 - p] + (PP_STORES if sy else 0) - (2 if dmac else 0)
                    if n is not None:
                        ev.append(("wait", n))
                ev.append(("bar",))
                if late:
                    dma2(1)                        # inside the MFMA cluster, after the barrier that opens it
                g += 1
        prev_full = full[j]
        if j + 1 < nt:
            # round 6: group 0's extra barrier in FRONT of the boundary refill, group 1's BEHIND its epilogue / publish (the two groups'
            # epilogues run side by side instead of one after the other); bad = "g1_barrier_first" is the round-5 order (also hazard-free:
            # the debug build keeps it as the A/B reference), "no_boundary_barrier" drops group 0's
            if (grp == 0 and bad != "no_boundary_barrier") or (grp == 1 and bad == "g1_barrier_first"):
                ev.append(("bar",))
            dma2()
            pre_issued = True
        else:
            prev_full = False
        for _ in range(stores[j]):
            ev.append(("vm", ("st", j)))
        if j in drain:                            # stream-K publish: s_waitcnt vmcnt(0) behind the slab stores, then the flag
            ev.append(("wait", 0))
        if j + 1 < nt and grp == 1 and bad != "g1_barrier_first":
            ev.append(("bar",))
    if grp == 0:
        ev.append(("bar",))
    return ev


SK_PUBLISH_STORES = 32                            # a wave's 128 x 64 fp32 block = 32 x 16 bytes per lane


@pytest.mark.parametrize("nk", [4, 5, 8, 16, 32])
def test_stream_k_item_sequences_schedule(nk):
    """gemm_bf16_mfma256sk_kernel (round 5) runs the v2 K-loop over ITEMS (nk K-tiles of one output tile) instead of whole tiles;
    per item the only thing that differs from a tile of the v2 kernel is what happens at its end:
      chained item   -- the next item continues in the same accumulators: no stores, prev_full = false;
      publish item   -- 32 write-through stores per wave (its fp32 block), then s_waitcnt vmcnt(0): prev_full = false;
      epilogue item  -- the normal fused epilogue: >= PP_STORES stores, prev_full = true  (a consume + epilogue item is always the
                        workgroup's last item: nothing follows it).
    RAW / WAR of every run shape the host plan produces, on the v2 program-order model: the M = 4096 QKV workgroups (publish,
    chain, epilogue | chain, epilogue, consume-epilogue), the fused residual projections (five chained items then publish /
    consume), long runs with several chained tiles."""
    import functools
    P, E, C = "publish", "epilogue", "chain"

    def run(kinds):
        nt = len(kinds)
        full = [k == E for k in kinds]
        stores = [PP_STORES if k == E else (SK_PUBLISH_STORES if k == P else 0) for k in kinds]
        drain = {j for j, k in enumerate(kinds) if k == P}
        pp_model(nk, nt, full, stores, stream=functools.partial(_pp2_stream, drain=drain))
        pp_model(nk, nt, full, stores, has_bias=False, stream=functools.partial(_pp2_stream, drain=drain))

    run([P, C, E])                     # QKV at M = 4096, even workgroup: half a tile published, then a whole tile in two chunks
    run([C, E, C])                     # odd workgroup up to its consume chain ...
    run([C, E, E])                     # ... whose last item ends with consume + epilogue (stores; nothing follows)
    run([C, C, C, C, P])               # fused residual projections at M = 4096: producer
    run([C, C, C, C, E])               # consumer
    run([C, C, C, C, E] * 4)           # a prefill chunk: four chained tiles of five chunks per workgroup
    run([P] + [C, C, C, E] * 2)        # B = 6144 QKV: publish first, then whole tiles of four chunks
    run([C] * 9 + [P] + [C] * 4 + [E])  # 20-chunk tiles in runs of 15: ten chunks published, five consumed
    run([P])
    run([E])


@pytest.mark.parametrize("nk", [4, 5, 8, 32])
@pytest.mark.parametrize("nt", [1, 2, 3, 5])
def test_persistent_gemm_v2_schedule(nk, nt):
    """RAW / WAR of the second-generation kernel's literal program order (both wave groups, the bias-issuing wave): interior
    tiles, more stores than assumed, edge tiles (no store-count waits), mixed."""
    pp_model(nk, nt, [True] * nt, [PP_STORES] * nt, stream=_pp2_stream)
    pp_model(nk, nt, [True] * nt, [2 * PP_STORES] * nt, stream=_pp2_stream)
    pp_model(nk, nt, [False] * nt, [0] * nt, stream=_pp2_stream)
    pp_model(nk, nt, [False] * nt, [PP_STORES] * nt, stream=_pp2_stream)
    pp_model(nk, nt, [j % 2 == 0 for j in range(nt)], [PP_STORES if j % 2 == 0 else 3 for j in range(nt)], has_bias=False,
             stream=_pp2_stream)


@pytest.mark.parametrize("nk", [8, 9, 32])
@pytest.mark.parametrize("nt", [1, 2, 3])
def test_persistent_gemm_v2_late_half_tile_experiment_schedule(nk, nt):
    """Experiment variant 26 (ABL 32768): in fast K-tiles the second half-tile of a phase is issued inside the MFMA cluster and
    group 1 waits for two instructions fewer -- hazards of that program order, mixed with generic K-tiles."""
    import functools
    st = functools.partial(_pp2_stream, dmac="fast")
    pp_model(nk, nt, [True] * nt, [PP_STORES] * nt, stream=st)
    pp_model(nk, nt, [j % 2 == 0 for j in range(nt)], [PP_STORES if j % 2 == 0 else 3 for j in range(nt)], has_bias=False, stream=st)


def test_persistent_v2_model_detects_errors():
    import functools
    with pytest.raises(AssertionError, match="RAW"):      # fewer stores than the waits assume
        pp_model(8, 3, [True] * 3, [PP_STORES - 8] * 3, stream=_pp2_stream)
    with pytest.raises(AssertionError, match="RAW"):      # store-count wait kept one phase too long (K-tile 2's loads are younger)
        pp_model(8, 3, [True] * 3, [PP_STORES] * 3, stream=functools.partial(_pp2_stream, bad="late_store_wait"))
    with pytest.raises(AssertionError):                   # boundary refill without its barrier: barrier counts / WAR break
        pp_model(8, 3, [True] * 3, [PP_STORES] * 3, stream=functools.partial(_pp2_stream, bad="no_boundary_barrier"))
    # the round-5 boundary (group 1's barrier in front of its refill: variant 94 of the debug build) is hazard-free too
    for nk in (4, 5, 8, 32):
        pp_model(nk, 3, [True] * 3, [PP_STORES] * 3, stream=functools.partial(_pp2_stream, bad="g1_barrier_first"))


def _pp2_ktile_walk(nk, nt, full):
    """Index bookkeeping of gemm_bf16_mfma256pp2_kernel's K loop, restated: the phase counter g (ring stage of compute K-tile =
    (g >> 1) & 1), the load cursor (tile lj, K-tile lt, running K-tile index lx -> ring stage lx & 1), the fast K-tiles of full
    tiles (static ring stage, unconditional issue, no wrap test in their cursor step).  Returns per-K-tile records
    (j, t, kind) with kind in {"generic", "fast"} and asserts what the fast path takes for granted."""
    g = 0
    lx, lt, lj = 0, 0, 0
    issued = []                                   # (tile, K-tile, half-tile kind, ring stage) in issue order

    def issue(kinds, ring=None, fast_expect=None):
        if lj >= nt:
            return
        r = lx & 1
        if ring is not None:
            assert ring == r, "static ring stage of a fast K-tile differs from the cursor's"
        if fast_expect is not None:
            assert (lj, lt) == fast_expect, ("fast K-tile issues outside its own tile", (lj, lt), fast_expect)
        for k in kinds:
            issued.append((lj, lt, k, r))

    def advance(fast=False):
        nonlocal lx, lt, lj
        lx += 1
        lt += 1
        if fast:
            assert lt < nk, "cursor of a fast K-tile reached the end of the tile"
        elif lt == nk:
            lt = 0
            lj += 1

    issue([0, 1, 2, 3])
    advance()
    issue([0, 1])
    walked = []
    pre_issued = False
    for j in range(nt):
        t = 0

        def generic(t):
            nonlocal g, pre_issued
            for p in (0, 1):
                if p == 0:
                    if t == 0 and pre_issued:
                        pre_issued = False
                    else:
                        issue([2, 3])
                        advance()
                else:
                    issue([0, 1])
                g += 1
            walked.append((j, t, "generic"))

        def fast(t, st):
            nonlocal g
            assert st == (g >> 1) & 1, "static compute stage differs from the phase counter's"
            issue([2, 3], ring=1 - st, fast_expect=(j, t + 1))
            advance(fast=True)
            g += 1
            assert st == (g >> 1) & 1
            issue([0, 1], ring=st, fast_expect=(j, t + 2))
            g += 1
            walked.append((j, t, "fast"))

        generic(0)
        generic(1)
        t = 2
        if full[j]:
            if (g & 2) and t <= nk - 3:
                generic(t)
                t += 1
            while t + 1 <= nk - 3:
                fast(t, 0)
                fast(t + 1, 1)
                t += 2
        while t < nk:
            generic(t)
            t += 1
        if j + 1 < nt:                            # tile boundary: W1 / A1 of the next tile's K-tile 1 ahead of the stores
            issue([2, 3])
            advance()
            pre_issued = True
    return walked, issued


@pytest.mark.parametrize("nk", [4, 5, 6, 7, 8, 9, 16, 31, 32, 33, 128])
@pytest.mark.parametrize("nt", [1, 2, 3, 4])
def test_persistent_gemm_v2_fast_ktiles_bookkeeping(nk, nt):
    """Every K-tile of every tile is computed exactly once and in order whichever mix of generic / fast K-tiles the loop
    takes; every half-tile of every K-tile is issued exactly once, in consumption order, into the ring stage its K-tile is
    read from; fast K-tiles never touch another tile and know their ring stages at compile time."""
    for full in ([True] * nt, [False] * nt, [j % 2 == 0 for j in range(nt)], [j % 2 == 1 for j in range(nt)]):
        walked, issued = _pp2_ktile_walk(nk, nt, full)
        assert [(j, t) for j, t, _ in walked] == [(j, t) for j in range(nt) for t in range(nk)]
        want = [(j, t, k) for j in range(nt) for t in range(nk) for k in range(4)]
        assert [(j, t, k) for j, t, k, _ in issued] == want
        for j, t, k, ring in issued:
            assert ring == (j * nk + t) & 1                      # K-tile (j, t) is compute K-tile number j * nk + t
        n_fast = sum(1 for _, _, kind in walked if kind == "fast")
        for j in range(nt):
            mine = [kind for jj, _, kind in walked if jj == j]
            if not full[j]:
                assert "fast" not in mine
            elif nk >= 8:
                assert mine.count("fast") >= nk - 6           # at most K-tiles 0, 1, one stage-alignment K-tile and the last 2-3
        assert n_fast % 2 == 0


# ---- one-tile 256 x 128 kernel (gemm_bf16_mfma256x128_kernel): three half-tiles per K-tile, ring of six slots ---------------------
T21_WAIT = int(re.search(r"constexpr int T21_WAIT = (\d+);", SRC).group(1))


def _t21_stream(grp, nk, bad=None):
    """program order of one wave: loads in consumption order A0 = 3t, W = 3t + 1, A1 = 3t + 2; two DMA instructions each"""
    ev = []

    def dma(sidx):
        ev.append(("vm", ("ld", sidx)))
        ev.append(("vm", ("ld", sidx)))

    def wait(steady):
        ev.append(("wait", T21_WAIT if steady else 0))

    for sidx in (0, 1, 2):
        dma(sidx)
    if nk > 1:
        dma(3)
        dma(4)
    if grp == 1:
        wait(nk >= 3)
        ev.append(("bar",))
    for t in range(nk):
        steady = t + 3 <= nk
        if bad == "always_steady":
            steady = True
        if grp == 0:
            wait(steady)
        ev.append(("bar",))
        ev.append(("read", ("ld", 3 * t)))
        ev.append(("read", ("ld", 3 * t + 1)))
        if t + 1 < nk:
            dma(3 * (t + 1) + 2)
        if grp == 1:
            wait(steady)
        ev.append(("bar",))
        if grp == 0:
            wait(steady)
        ev.append(("bar",))
        ev.append(("read", ("ld", 3 * t + 2)))
        if t + 2 < nk:
            dma(3 * (t + 2))
            dma(3 * (t + 2) + 1)
        if grp == 1:
            wait(t + 4 <= nk or bad == "always_steady")
        ev.append(("bar",))
    if grp == 0:
        ev.append(("bar",))
    return ev


def t21_model(nk, bad=None):
    waves = {g: _pp_analyse(_t21_stream(g, nk, bad)) for g in (0, 1)}
    assert len({v[3] for v in waves.values()}) == 1, "barrier counts differ between wave groups"
    last_read = {}
    for key, (issue, done, reads, _) in waves.items():
        for tag, b in reads:
            last_read[tag] = max(last_read.get(tag, -1), b)
            for okey, (oissue, odone, _, _) in waves.items():
                assert tag in odone and odone[tag] <= b - 1, f"RAW nk={nk}: {tag} read by group {key} not waited by group {okey}"
    for key, (issue, done, reads, _) in waves.items():
        for tag, b in list(issue.items()):
            if tag[0] == "ld" and tag[1] >= 6:
                assert b >= last_read[("ld", tag[1] - 6)] + 1, f"WAR nk={nk}: {tag} issued by group {key} too early"


@pytest.mark.parametrize("nk", [1, 2, 3, 4, 5, 8, 32, 128])
def test_one_tile_256x128_schedule(nk):
    t21_model(nk)


def test_one_tile_256x128_model_detects_a_missing_drain():
    with pytest.raises(AssertionError, match="RAW"):
        t21_model(8, bad="always_steady")


# ---- 256 x 128 slab kernel of the mid-batch stacks (gemm_bf16_mfma256x128_slab_kernel): A streamed by group 0 (two K-tiles deep), W by
# group 1 (TS_WD K-tiles deep): each group has its own vmcnt queue ----------------------------------------------------------------
TS_WD = int(re.search(r"constexpr int TS_WD = (\d+);", SRC).group(1))
_SLAB = SRC[SRC.index("void gemm_bf16_mfma256x128_slab_kernel("):]
# (the K-loop ends where the round-6 fused form -- hand-off + epilogue, its own workgroup barriers -- or the slab stores begin)
_SLAB = re.sub(r"\s+", "", re.sub(r"//[^\n]*", "", _SLAB[:min(_SLAB.index("// ---- slab stores"), _SLAB.index("const EpiArgs& ep = tf.ep[side];"))]))
# the lines the streams below restate (comments and whitespace removed)
for piece in ("if(!grp1){if(half_mode){for(intkt=0;kt<3&&kt<nk;++kt)issue_ah(kt);}else{issue_a(I0{},0);issue_a(I1{},0);if(nk>1)issue_a(I0{},1);}}"
              "else{for(intkt=0;kt<TS_WD&&kt<nk;++kt){if(half_mode)issue_wh(kt);elseissue_w(kt);}}",
              "if(grp1){ts_wait_tiles(min(TS_WD,nk)-1);cbarrier();}",
              "if(!grp1)ts_wait_tiles(t+1<nk?2:1);cbarrier();",
              "if(!grp1&&t+1<nk)issue_a(I1{},t+1);if(grp1)asmvolatile(\"s_waitcntlgkmcnt(0)\":::\"memory\");cbarrier();",
              "if(!grp1)ts_wait_tiles(t+1<nk?2:0);cbarrier();",
              "if(!grp1&&t+2<nk)issue_a(I0{},t+2);if(grp1){if(t+TS_WD<nk)issue_w(t+TS_WD);ts_wait_tiles(min(t+TS_WD,nk-1)-(t+1));"
              "asmvolatile(\"s_waitcntlgkmcnt(0)\":::\"memory\");}cbarrier();",
              # half mode (round 6): one phase per K-tile (two barriers), A ring of four slots
              "constuint32_tring=a_lds+(uint32_t)(kt&3)*(uint32_t)HALF_BYTES;",
              "constchar*abuf=smem+(t&3)*HALF_BYTES;",
              "if(!grp1)ts_wait_tiles(min(t+2,nk-1)-t);cbarrier();",
              "if(!grp1&&t+3<nk)issue_ah(t+3);if(grp1){ts_wait_tiles(max(min(t+TS_WD-1,nk-1)-(t+1),0));"
              "asmvolatile(\"s_waitcntlgkmcnt(0)\":::\"memory\");}cbarrier();if(grp1&&t+TS_WD<nk)issue_wh(t+TS_WD);",
              "TS_STAMP(6);if(!grp1)cbarrier();TS_STAMP(7);"):
    assert piece in _SLAB, piece
assert _SLAB.count("cbarrier();") == 8 and _SLAB.count("ts_wait_tiles(") == 6
assert "if(tiles>=4)wait_vmcnt_n<16>();elseif(tiles==3)wait_vmcnt_n<12>();elseif(tiles==2)wait_vmcnt_n<8>();elseif(tiles==1)wait_vmcnt_n<4>();elsewait_vmcnt_n<0>();" \
    in re.sub(r"\s+", "", SRC[SRC.index("__device__ inline void ts_wait_tiles(int tiles)"):][:400])


def _slab_stream(grp, nk, bad=None, half=False):
    """program order of one wave; a tile (A half-tile or W) is four DMA instructions of the issuing wave.  half: the round-6 fused form for
    129 .. 256 rows -- a workgroup owns one 128-row half of a column tile: ONE phase per K-tile (two barriers), A ring of four slots"""
    ev = []

    def dma(tag):
        for _ in range(4):
            ev.append(("vm", tag))

    def wait_tiles(n):
        ev.append(("wait", 4 * min(max(n, 0), 4)))

    if half:     # one phase per K-tile (two barriers), group 1 one barrier behind
        if grp == 0:
            for kt in range(min(3, nk)):
                dma(("a0", kt))
        else:
            for kt in range(min(TS_WD, nk)):
                dma(("w", kt))
            wait_tiles(min(TS_WD, nk) - 1)
            ev.append(("bar",))
        for t in range(nk):
            if grp == 0:
                wait_tiles(min(t + 2, nk - 1) - t + (1 if bad == "half_a0_late" else 0))
            ev.append(("bar",))
            ev.append(("read", ("w", t)))
            ev.append(("read", ("a0", t)))
            if grp == 0 and t + 3 < nk:
                dma(("a0", t + 3))
            if grp == 1:
                wait_tiles(max(min(t + TS_WD - 1, nk - 1) - (t + 1), 0) + (1 if bad == "half_w_late" else 0))
            if grp == 1 and bad == "half_w_early" and t + TS_WD < nk:
                dma(("w", t + TS_WD))
            ev.append(("bar",))
            if grp == 1 and bad != "half_w_early" and t + TS_WD < nk:
                dma(("w", t + TS_WD))
        if grp == 0:
            ev.append(("bar",))
        return ev
    if grp == 0:
        dma(("a0", 0))
        dma(("a1", 0))
        if nk > 1:
            dma(("a0", 1))
    else:
        for kt in range(min(TS_WD, nk)):
            dma(("w", kt))
        wait_tiles(min(TS_WD, nk) - 1)
        ev.append(("bar",))
    for t in range(nk):
        if grp == 0:
            wait_tiles(2 if t + 1 < nk else 1)
        ev.append(("bar",))
        ev.append(("read", ("w", t)))
        ev.append(("read", ("a0", t)))
        if grp == 0 and t + 1 < nk:
            dma(("a1", t + 1))
        ev.append(("bar",))
        if grp == 0:
            wait_tiles((2 if t + 1 < nk else 0) + (1 if bad == "a1_late" else 0))
        ev.append(("bar",))
        ev.append(("read", ("a1", t)))
        if grp == 0 and t + 2 < nk:
            dma(("a0", t + 2))
        if grp == 1:
            if t + TS_WD < nk:
                dma(("w", t + TS_WD))
            wait_tiles(min(t + TS_WD, nk - 1) - (t + 1) + (1 if bad == "w_late" else 0))
        ev.append(("bar",))
    if grp == 0:
        ev.append(("bar",))
    return ev


def slab_model(nk, bad=None, depth=None, half=False):
    waves = {g: _pp_analyse(_slab_stream(g, nk, bad, half)) for g in (0, 1)}
    assert len({v[3] for v in waves.values()}) == 1, "barrier counts differ between wave groups"
    owner = {"a0": 0, "a1": 0, "w": 1}
    ring = {"a0": 4 if half else 2, "a1": 2, "w": depth or TS_WD}
    last_read = {}
    for key, (issue, done, reads, _) in waves.items():
        for tag, b in reads:
            last_read[tag] = max(last_read.get(tag, -1), b)
            odone = waves[owner[tag[0]]][1]
            assert tag in odone and odone[tag] <= b - 1, f"RAW nk={nk}: {tag} read by group {key} before its issuer waited"
    for key, (issue, done, reads, _) in waves.items():
        for tag, b in list(issue.items()):
            if tag[0] in ring and tag[1] >= ring[tag[0]]:
                prev = (tag[0], tag[1] - ring[tag[0]])
                assert b >= last_read[prev] + 1, f"WAR nk={nk}: {tag} issued by group {key} too early"
    # every tile is issued exactly once, by its owner, and read by both groups
    for kind in ring:
        if half and kind == "a1":
            assert not any(t[0] == "a1" for t in waves[0][0])
            continue
        assert sorted(t[1] for t in waves[owner[kind]][0] if t[0] == kind) == list(range(nk))
        for g in (0, 1):
            assert sorted(t[1] for t, _ in waves[g][2] if t[0] == kind) == list(range(nk))


@pytest.mark.parametrize("nk", [1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 16, 32])
def test_slab_kernel_schedule(nk):
    slab_model(nk)
    slab_model(nk, half=True)          # round 6: one 128-row half per workgroup (fused epilogues, 129 .. 256 rows)


def test_slab_kernel_half_mode_model_detects_errors():
    with pytest.raises(AssertionError, match="RAW"):
        slab_model(8, bad="half_a0_late", half=True)
    with pytest.raises(AssertionError, match="RAW"):
        slab_model(8, bad="half_w_late", half=True)
    with pytest.raises(AssertionError, match="WAR"):
        slab_model(12, bad="half_w_early", half=True)     # W(t + TS_WD) in front of the barrier that closes group 1's reads of t


def test_slab_kernel_model_detects_errors():
    with pytest.raises(AssertionError, match="RAW"):
        slab_model(8, bad="w_late")              # one more W tile left in flight than was issued behind the awaited one
    with pytest.raises(AssertionError, match="RAW"):
        slab_model(8, bad="a1_late")
    with pytest.raises(AssertionError, match="WAR"):
        slab_model(12, depth=TS_WD - 1)          # a ring one slot shorter than the issue distance
