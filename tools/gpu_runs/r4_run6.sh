#!/bin/bash
# round 4, GPU call 6: 5..16-row stacks on the (now 4-launch) slab path vs the MFMA weight-streamer launches, same box, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python tools/ab_small_batch.py --knob few_max4 --batches 3,5,6,8,12,16 --reps 2 > $O/r4_ab_few_max4.jsonl 2> $O/r4_ab_few_max4.log
cat $O/r4_ab_few_max4.jsonl; tail -3 $O/r4_ab_few_max4.log
