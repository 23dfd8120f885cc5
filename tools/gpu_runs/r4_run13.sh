#!/bin/bash
# round 4, GPU call 13: 128..256 rows, fused one-slice QKV / h->4h GEMMs on 64x64 vs 128x64 vs 128x128 tiles, same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python tools/ab_small_batch.py --knob mid_tile --cutoffs 0,1,2 --batches 128,192,256 --reps 2 > $O/r4_ab_mid_tile.jsonl 2> $O/r4_ab_mid_tile.log
cat $O/r4_ab_mid_tile.jsonl; tail -3 $O/r4_ab_mid_tile.log
