#!/bin/bash
# round 6, GPU call 6: where does the hand-off time-out of the P128/D2048 runs come from?  tools/handoff_diag.py (2-layer 1.2B dims,
# B = 4096): debug library with the never-cleared diag buffer, product library, short decode with many repetitions
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 300 python tools/handoff_diag.py --debug-lib --reps 3 > $O/r6_handoff_diag_dbg.log 2>&1; tail -25 $O/r6_handoff_diag_dbg.log
timeout 300 python tools/handoff_diag.py --reps 3 > $O/r6_handoff_diag_product.log 2>&1; tail -12 $O/r6_handoff_diag_product.log
timeout 300 python tools/handoff_diag.py --debug-lib --decode 128 --prefill 2048 --reps 30 > $O/r6_handoff_diag_short.log 2>&1; grep -c ok $O/r6_handoff_diag_short.log; grep -A12 ERROR $O/r6_handoff_diag_short.log | head -30
timeout 300 python tools/handoff_diag.py --debug-lib --no-graphs --reps 2 > $O/r6_handoff_diag_eager.log 2>&1; tail -12 $O/r6_handoff_diag_eager.log
