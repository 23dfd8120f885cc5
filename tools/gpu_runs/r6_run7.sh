#!/bin/bash
# round 6, GPU call 7: hand-off time-out diagnosis at full depth (12 + 12 layers), P128/D2048, B = 4096
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 400 python tools/handoff_diag.py --debug-lib --layers 12 --reps 3 > $O/r6_handoff_diag_full_dbg.log 2>&1; tail -25 $O/r6_handoff_diag_full_dbg.log
timeout 400 python tools/handoff_diag.py --layers 12 --reps 3 > $O/r6_handoff_diag_full_product.log 2>&1; tail -12 $O/r6_handoff_diag_full_product.log
