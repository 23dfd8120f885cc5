#!/bin/bash
# round 3, GPU call 16: one-tile 256 x 128 kernel: tests vs fp64 + A/B vs the 128^2 kernel and the vendor library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "256x128 or test_linear" ) 2>&1 | tail -4
timeout 200 python tools/kernel_probe.py t21 > $O/r3_t21.jsonl 2> $O/r3_t21.err
cat $O/r3_t21.jsonl; tail -2 $O/r3_t21.err
