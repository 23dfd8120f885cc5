#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "attention_local or attention_rows" ) 2>&1 | tail -3
python tools/kernel_probe.py local --batch 4096 --iters 20 > $O/r2_probe_local_new.jsonl 2> $O/r2_probe_local_new.err; cat $O/r2_probe_local_new.jsonl
