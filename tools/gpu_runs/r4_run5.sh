#!/bin/bash
# round 4, GPU call 5: mid-batch attention fold (tests + same-box A/B), few-rows tests after moving the fused GEMV attention to the debug build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests/test_model_gpu.py -q -m gpu -x -s -k "mid_batch or mid_rows or fused_attention or small_batch_fused or bf16_generate_tracks or streamer_stack" 2>&1 | grep -v "^$" | tail -40 > $O/r4_t5_model.log
timeout 500 python tools/ab_small_batch.py --knob mid_attn --batches 24,32,64 > $O/r4_ab_mid_attn.jsonl 2> $O/r4_ab_mid_attn.log
for b in 16 32 64; do
  timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-parity --no-secondary > $O/r4_bench_sm3_b$b.json 2> $O/r4_bench_sm3_b$b.log
done
cat $O/r4_t5_model.log; cat $O/r4_ab_mid_attn.jsonl; tail -3 $O/r4_ab_mid_attn.log
for b in 16 32 64; do python -c "import json;d=json.load(open('$O/r4_bench_sm3_b$b.json'));print('B=$b', round(d['value'],1), 'tok/s', 'decode', round(d.get('decode_phase_tokens_per_s_per_gpu',0)))"; done
