#!/bin/bash
# round 6, GPU call 2: (1) the hand-off time-out tests with a PUBLISHING workgroup dropped; (2) GELU rewrite (asm min / max, degree 11):
# accuracy + every GELU GEMM test, interleaved A/B against the round-5 form (variant 30); (3) QKV epilogue with the cos / sin rows of all
# eight fragments in one batch (variant 29) vs production; (4) the tile-boundary timeline of the persistent kernel (variant 28);
# (5) the default bench line on the new product library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "handoff or gelu or test_linear" > $O/r6_run2_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -3 $O/r6_run2_kernel_tests.log
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "handoff" > $O/r6_run2_model_tests.log 2>&1
echo "model tests rc=$?"; tail -3 $O/r6_run2_model_tests.log
timeout 300 python tools/kernel_probe.py pptl > $O/r6_pp_timeline.jsonl 2> $O/r6_pp_timeline.err
echo "pptl rc=$?"; tail -2 $O/r6_pp_timeline.err; python - <<'PY'
import json
for l in open('gpurun_out/r6_pp_timeline.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    if d.get('op') != 'pp_timeline': continue
    print(d['mode'], d['M'], d['N'], 'tiles', d['tiles_per_workgroup'], 'us', d['us_production'], d['us_with_stamps'], 'ns/tick', d['ns_per_tick_if_span_is_the_launch'])
    print('   med', {k.split(' ')[0]: v for k, v in d['median_ticks'].items()})
    print('   p90', {k.split(' ')[0]: v for k, v in d['p90_ticks'].items()})
    print('   epi by wave', d['epilogue_median_ticks_by_wave'])
PY
timeout 400 python tools/kernel_probe.py ppab --variant 12 --ab-other 30 > $O/r6_ppab_gelu_r6_vs_r5.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r6_ppab_gelu_r6_vs_r5.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    if d.get('op') == 'pp_ab' and d.get('mode') == 'gelu': print('gelu r6 (a) vs r5 (b)', d['M'], d['N'], d['K'], d['a_us'], d['b_us'], d['a_TF'], d['b_TF'], d['bit_identical'])
PY
timeout 300 python tools/kernel_probe.py ppqkv --variant 12 --ab-other 29 > $O/r6_ppqkv_rope_one_batch.jsonl 2>/dev/null
cut -c1-300 $O/r6_ppqkv_rope_one_batch.jsonl
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/r6_bench_gelu28.json
python -c "
import json; d=json.loads(open('$O/r6_bench_gelu28.json').read()); print('default', round(d['value'],1), d['ms_per_step'], (d.get('parity_in_run') or {}).get('ok'), d['roofline']['frac'], d.get('kernel_family_ms'))"
