#!/bin/bash
# round 4, GPU call 23 (final tree): the model tests that run long contexts at small batches (16-wave global attention) + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 280 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "full_size or first_block_logits or real_dims or long_prompt or long_decode or measure_generation or inference_demo or persistent_gemv" 2>&1 | tail -3
