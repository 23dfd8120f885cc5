#!/bin/bash
# round 4, GPU call 4: same-box A/Bs (fused local attention, dual mid-batch GEMM, t21 lgkmcnt guard), B = 128 / 256 after the slab-only fix,
# compaction speed-up with the C-ABI kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 400 python tools/ab_small_batch.py --knob few_attn --batches 1,2 > $O/r4_ab_few_attn.jsonl 2> $O/r4_ab_few_attn.log
timeout 400 python tools/ab_small_batch.py --knob mid_dual --batches 32,64 > $O/r4_ab_mid_dual.jsonl 2> $O/r4_ab_mid_dual.log
timeout 300 python tools/kernel_probe.py t21 > $O/r4_t21_safe_ab.jsonl 2>&1
for b in 128 256; do
  timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-parity --no-secondary > $O/r4_bench_sm2_b$b.json 2> $O/r4_bench_sm2_b$b.log
done
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -x -s -k "compaction_speeds or mid_batch_dual" 2>&1 | grep -v "^$" | tail -14 > $O/r4_t4_model.log
cat $O/r4_ab_few_attn.jsonl $O/r4_ab_mid_dual.jsonl; tail -3 $O/r4_ab_few_attn.log
python - <<'P'
import json
for l in open('gpurun_out/r4_t21_safe_ab.jsonl'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['M'],d['N'],d['K'],d['mode'],d['TF'])
P
for b in 128 256; do python -c "import json;d=json.load(open('$O/r4_bench_sm2_b$b.json'));print('B=$b', round(d['value'],1), 'tok/s', 'decode', round(d.get('decode_phase_tokens_per_s_per_gpu',0)))"; done
cat $O/r4_t4_model.log
