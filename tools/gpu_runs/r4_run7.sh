#!/bin/bash
# round 4, GPU call 7: where should the GEMV path (<= N rows) hand over to the slab path?  cutoffs 16 / 4 / 2 / 1 / 0 at B = 1..4, same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python tools/ab_small_batch.py --knob few_max4 --cutoffs 16,4,2,1,0 --batches 1,2,3,4 --reps 2 > $O/r4_ab_few_cutoffs.jsonl 2> $O/r4_ab_few_cutoffs.log
cat $O/r4_ab_few_cutoffs.jsonl; tail -3 $O/r4_ab_few_cutoffs.log
