#!/bin/bash
# round 4, GPU call 20: 16-wave single-launch global attention for tiny batches: kernel tests vs fp64, model tests, same-box A/B at B = 1 / 2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "attention or attn" 2>&1 | tail -3
timeout 400 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "first_block_logits_vs_oracle or small_batch or unwritten_cache or long_decode_properties or persistent_gemv or p2048_d2048" 2>&1 | tail -3
timeout 300 python tools/ab_small_batch.py --knob attn_wide --batches 1,2 --reps 4 > $O/r4_ab_attn_wide.jsonl 2> $O/r4_ab_attn_wide.log
cat $O/r4_ab_attn_wide.jsonl; tail -2 $O/r4_ab_attn_wide.log
