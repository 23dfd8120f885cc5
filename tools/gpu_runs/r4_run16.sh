#!/bin/bash
# round 4, GPU call 16: 96..256 rows: the 4-launch structure (dual GEMM with two forced K slices + merged reduce + attention fold) vs the
# separate fused-epilogue launches, same box, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python tools/ab_small_batch.py --knob mid_force2 --batches 96,128,192,256 --reps 3 > $O/r4_ab_mid_force2.jsonl 2> $O/r4_ab_mid_force2.log
cat $O/r4_ab_mid_force2.jsonl; tail -3 $O/r4_ab_mid_force2.log
