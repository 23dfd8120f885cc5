#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x -k "keeping_logits" ) 2>&1 | tail -3
bash tools/gpu_runs/r3_pmc.sh a0867d0
