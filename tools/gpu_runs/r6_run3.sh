#!/bin/bash
# round 6, GPU call 3: the tile boundary with the two wave groups' epilogues side by side (group 1's extra barrier behind its epilogue):
# correctness of every persistent / stream-K path, timeline old vs new boundary, interleaved A/B (variant 31 = round-5 boundary),
# stream-K probe, default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "linear or stream_k or qkv or logits_argmax or gelu" > $O/r6_run3_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -3 $O/r6_run3_kernel_tests.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "g6_g9 or full_depth_1p2b_bf16_first_block or benchmark_batch or handoff" > $O/r6_run3_model_tests.log 2>&1
echo "model tests rc=$?"; tail -3 $O/r6_run3_model_tests.log
for v in 28 32; do
timeout 300 python tools/kernel_probe.py pptl --variant $v > $O/r6_pp_timeline_v$v.jsonl 2> /dev/null
python - <<PY
import json
for l in open('gpurun_out/r6_pp_timeline_v$v.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    if d.get('op') != 'pp_timeline': continue
    print(d['boundary'], d['mode'], d['M'], d['N'], 'tiles', d['tiles_per_workgroup'], 'us', d['us_production'], d['us_with_stamps'])
    print('   med', {k.split(' ')[0]: v for k, v in d['median_ticks'].items()})
    print('   g0', {k.split(' ')[0]: v for k, v in d['group0_median_ticks'].items()}, 'g1', {k.split(' ')[0]: v for k, v in d['group1_median_ticks'].items()})
PY
done
timeout 500 python tools/kernel_probe.py ppab --variant 12 --ab-other 31 > $O/r6_ppab_boundary_r6_vs_r5.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r6_ppab_boundary_r6_vs_r5.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    if d.get('op') == 'pp_ab': print('boundary r6 (a) vs r5 (b)', d['mode'], d['M'], d['N'], d['K'], d['a_us'], d['b_us'], d['a_TF'], d['b_TF'], d['bit_identical'])
PY
timeout 300 python tools/kernel_probe.py ppqkv --variant 12 --ab-other 31 > $O/r6_ppqkv_boundary_r6_vs_r5.jsonl 2>/dev/null
cut -c1-300 $O/r6_ppqkv_boundary_r6_vs_r5.jsonl
timeout 300 python tools/kernel_probe.py sk > $O/r6_decode_shapes_sk.jsonl 2>/dev/null
cut -c1-400 $O/r6_decode_shapes_sk.jsonl
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/r6_bench_boundary.json
python -c "
import json; d=json.loads(open('$O/r6_bench_boundary.json').read()); print('default', round(d['value'],1), d['ms_per_step'], (d.get('parity_in_run') or {}).get('ok'), d['roofline']['frac'], d.get('kernel_family_ms'))"
