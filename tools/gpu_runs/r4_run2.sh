#!/bin/bash
# round 4, GPU call 2: guard test fix, fused local attention (bit-identity + small-batch stacks), B = 1 / 2 / 4 bench A/B, default bench with secondary
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -s -k "device_guard or fused_attention or small_batch_fused or few_rows_dual or token_decoder_generate or generate_greedy" 2>&1 | grep -v "^$" | tail -25 > $O/r4_t2_model.log
for b in 1 2 4; do
  timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-parity --no-secondary > $O/r4_bench_fuseattn_b$b.json 2> $O/r4_bench_fuseattn_b$b.log
done
timeout 900 python bench.py --steps 3 --warmup 1 > $O/r4_bench_second.json 2> $O/r4_bench_second.log
echo "bench rc $?" >> $O/r4_bench_second.log
cat $O/r4_t2_model.log
for b in 1 2 4; do python -c "import json;d=json.load(open('$O/r4_bench_fuseattn_b$b.json'));print('B=$b', round(d['value'],1), 'tok/s', round(d['ms_per_step'],2),'ms')"; done
tail -12 $O/r4_bench_second.log
python - <<'P'
import json
d=json.load(open('gpurun_out/r4_bench_second.json'))
print('value', d['value'], 'ms', d['ms_per_step'])
print('parity', {k:v for k,v in d['parity_in_run'].items() if k!='what'})
for s in d.get('secondary',[]): print(s['workload'], s['value'], s['roofline_decode_phase']['frac_of_binding_roof'])
print('roofline', d['roofline']['frac'], d['kernel_family_ms'])
P
