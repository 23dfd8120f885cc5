#!/bin/bash
# round 4, GPU call 15 (final tree): smoke, the whole -m gpu suite, the default bench exactly as the driver runs it, configs[3] per GPU,
# the harness counterpart at fixed batches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 1500 python -m pytest tests/ -q -m gpu --durations=15 ) > $O/r4_pytest_full_final.log 2>&1
grep -v "^\.\|^$" $O/r4_pytest_full_final.log | tail -28 > $O/r4_pytest_gpu_final_tail.txt; cat $O/r4_pytest_gpu_final_tail.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r4_bench_default_final.json 2> $O/r4_bench_default_final.log
tail -4 $O/r4_bench_default_final.log
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r4_bench_default_final.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'families', d['kernel_family_ms'])
print('parity', {k:v for k,v in d['parity_in_run'].items() if k not in ('what',)})
for s in d.get('secondary',[]): print(s['workload'][:40], s['value'], s['roofline_decode_phase']['frac_of_binding_roof'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
P
timeout 400 python bench.py --prefill 2048 --decode 2048 --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --no-roofline > $O/r4_bench_p2048_d2048.json 2> $O/r4_bench_p2048_d2048.log
python -c "import json;d=json.load(open('$O/r4_bench_p2048_d2048.json'));print('P2048/D2048 B', d['config']['batch_per_gpu'], round(d['value'],1), 'tok/s')"
mkdir -p $O/r4_results
timeout 600 python -m block_transformer_amd.measure_generation_time --config-name block_main_b4_1.2b "++benchmark_batch_sizes=[1,8,64,4096]" ++benchmark_prefill_length=2048 ++benchmark_decode_length=128 ++save_dir=$O/r4_results > $O/r4_results/generation_time.log 2>&1
tail -12 $O/r4_results/generation_time.log
