#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "flash" ) 2>&1 | tail -4
timeout 300 python tools/kernel_probe.py flash --batch 256 --keys 512 --iters 10 > $O/r2_probe_flash_xcd2.jsonl 2> $O/r2_probe_flash_xcd2.err; cat $O/r2_probe_flash_xcd2.jsonl; tail -3 $O/r2_probe_flash_xcd2.err
