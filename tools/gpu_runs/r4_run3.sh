#!/bin/bash
# round 4, GPU call 3: pipelined fused local attention (B = 1 / 2), DPP butterfly in local_attend_kernel, dual mid-batch GEMM (B = 32 .. 256)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests/test_model_gpu.py -q -m gpu -x -s -k "fused_attention or small_batch_fused or few_rows_dual or mid_batch or mid_rows or token_decoder_generate or bf16_generate_tracks" 2>&1 | grep -v "^$" | tail -30 > $O/r4_t3_model.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "attention or local or attn" 2>&1 | tail -4 > $O/r4_t3_kernels.log
timeout 200 python tools/kernel_probe.py local --batch 4096 > $O/r4_probe_local_dpp.jsonl 2>&1
for b in 1 2 32 64 128 256; do
  timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-parity --no-secondary > $O/r4_bench_sm_b$b.json 2> $O/r4_bench_sm_b$b.log
done
cat $O/r4_t3_model.log; cat $O/r4_t3_kernels.log; cat $O/r4_probe_local_dpp.jsonl | tail -8
for b in 1 2 32 64 128 256; do python -c "import json;d=json.load(open('$O/r4_bench_sm_b$b.json'));print('B=$b', round(d['value'],1), 'tok/s', round(d['ms_per_step'],2),'ms', 'decode', round(d.get('decode_phase_tokens_per_s_per_gpu',0)))"; done
