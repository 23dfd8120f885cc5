#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "flash" ) 2>&1 | tail -4
for cfg in "256 512" "64 2048" "1024 128" "512 256" "64 512"; do
  set -- $cfg
  timeout 100 python tools/kernel_probe.py flash --batch $1 --keys $2 --iters 10 2>/dev/null | tail -2
done > $O/r2_probe_flash_v3b.jsonl
cat $O/r2_probe_flash_v3b.jsonl
