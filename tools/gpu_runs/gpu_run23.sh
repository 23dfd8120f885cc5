#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
for cfg in "256 512" "64 2048" "16 8192" "1024 128" "512 256"; do
  set -- $cfg
  timeout 200 python tools/kernel_probe.py flash --batch $1 --keys $2 --iters 10 --split 2 2>/dev/null | grep -E "production|round-1" | tail -2
done > $O/r2_probe_flash_lengths.jsonl
cat $O/r2_probe_flash_lengths.jsonl
