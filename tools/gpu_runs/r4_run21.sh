#!/bin/bash
# round 4, GPU call 21: the two batch-1 / 2 changes re-measured with alternating order and 6 repetitions (best-of per setting)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 300 python tools/ab_small_batch.py --knob few_persist --batches 1,2 --reps 6 > $O/r4_ab_few_persist_v2.jsonl 2> $O/r4_ab_few_persist_v2.log
timeout 300 python tools/ab_small_batch.py --knob attn_wide --batches 1,2 --reps 6 > $O/r4_ab_attn_wide_v2.jsonl 2> $O/r4_ab_attn_wide_v2.log
cat $O/r4_ab_few_persist_v2.jsonl $O/r4_ab_attn_wide_v2.jsonl
