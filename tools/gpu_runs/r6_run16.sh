#!/bin/bash
# round 6, GPU call 16: is the QKV epilogue's store phase slow because of WHERE k / v go?  variant 98 stores them into the qkv buffer (timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 300 python tools/kernel_probe.py ppqkv --variant 12 --ab-other 98 > $O/r6_ppqkv_kv_stores_into_out_timing.jsonl 2>/dev/null
cut -c1-300 $O/r6_ppqkv_kv_stores_into_out_timing.jsonl
