#!/bin/bash
# round 4, GPU call 22: one setting per process, 40 generates each, processes alternated (on, off) x 3: persistent GEMV and 16-wave attention at B = 1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -f $O/r4_ab_b1_processes.jsonl
for rep in 1 2 3; do
  for knob in few_persist attn_wide; do
    for v in 1 0; do
      timeout 100 python tools/ab_small_batch.py --knob $knob --fixed $v --batches 1 --iters 40 >> $O/r4_ab_b1_processes.jsonl 2>> $O/r4_ab_b1_processes.log
    done
  done
done
cat $O/r4_ab_b1_processes.jsonl
