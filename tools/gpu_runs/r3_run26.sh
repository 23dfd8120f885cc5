#!/bin/bash
# round 3, GPU call 26: fused-QKV mode of the persistent kernel: production vs round-2 priority setting (24) vs generic loop (18)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for V in 24 18 24; do
  timeout 300 python tools/kernel_probe.py ppqkv --variant 12 --ab-other $V >> $O/r3_ppqkv_ab.jsonl 2> $O/r3_ppqkv_ab.err
done
cut -c1-330 $O/r3_ppqkv_ab.jsonl; tail -2 $O/r3_ppqkv_ab.err
