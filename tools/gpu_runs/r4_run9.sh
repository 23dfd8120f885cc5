#!/bin/bash
# round 4, GPU call 9: NaN-poisoned caches on every batch-size path; the rest of the model suite after the fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "unwritten_cache or keeping_logits" 2>&1 | tail -15 > $O/r4_t9_a.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu 2>&1 | tail -15 > $O/r4_t9_model.log
cat $O/r4_t9_a.log $O/r4_t9_model.log
