#!/bin/bash
# round 6, GPU call 5: (1) group 0's phase-(0,0) init + reads in front of the first barrier (+ the one-batch RoPE loads as default):
# correctness of the persistent / stream-K paths, interleaved A/B against variant 96 (barrier move only); (2) timeline with the stamps
# inside the epilogue and kernel entry / exit; (3) P128/D2048 in-situ A/B of the tile boundary; (4) compaction ratio at EOS-heavy
# B = 1024; (5) default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "linear or stream_k or qkv or logits_argmax or gelu" > $O/r6_run5_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -3 $O/r6_run5_kernel_tests.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "g6_g9 or full_depth_1p2b_bf16_first_block or benchmark_batch or handoff or steady_state" > $O/r6_run5_model_tests.log 2>&1
echo "model tests rc=$?"; tail -3 $O/r6_run5_model_tests.log
timeout 500 python tools/kernel_probe.py ppab --variant 12 --ab-other 96 > $O/r6_ppab_early_reads_vs_not.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r6_ppab_early_reads_vs_not.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    if d.get('op') == 'pp_ab': print('early reads (a) vs not (b)', d['mode'], d['M'], d['N'], d['K'], d['a_us'], d['b_us'], d['a_TF'], d['b_TF'], d['bit_identical'])
PY
timeout 300 python tools/kernel_probe.py ppqkv --variant 12 --ab-other 96 > $O/r6_ppqkv_early_reads_vs_not.jsonl 2>/dev/null
cut -c1-300 $O/r6_ppqkv_early_reads_vs_not.jsonl
timeout 300 python tools/kernel_probe.py pptl > $O/r6_pp_timeline_final.jsonl 2> $O/r6_pp_timeline_final.err
tail -2 $O/r6_pp_timeline_final.err
python - <<'PY'
import json
for l in open('gpurun_out/r6_pp_timeline_final.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    if d.get('op') != 'pp_timeline': continue
    m = d['median_ticks']
    print(d['mode'], d['M'], d['N'], 'tiles', d['tiles_per_workgroup'], 'us', d['us_production'], d['us_with_stamps'], 'kernel cyc', d.get('kernel_cycles_median'), 'fill', d.get('fill_cycles_entry_to_first_phase_median'),
          'tail', d.get('tail_cycles_last_epilogue_end_to_exit_median'), 'last epi', d.get('last_epilogue_cycles_median'), d.get('realtime_us'))
    print('   ', {k.split(' ')[0]: v for k, v in m.items()})
    print('    g0', list(d['group0_median_ticks'].values()), 'g1', list(d['group1_median_ticks'].values()))
PY
for v in 1 0; do
  timeout 600 python tools/bench_knobs.py serial_epilogues=$v -- --prefill 128 --decode 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings > $O/r6_ab_serial_epi${v}_p128.json 2> $O/r6_ab_serial_epi${v}_p128.err
  tail -c 300 $O/r6_ab_serial_epi${v}_p128.json; echo; tail -2 $O/r6_ab_serial_epi${v}_p128.err
done
timeout 600 python -m pytest tests/test_model_gpu.py -q -s -m gpu -k "compaction_speeds_up" > $O/r6_compaction_ratio.log 2>&1
grep -i "compaction\|ratio\|passed\|failed" $O/r6_compaction_ratio.log | tail -8
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/r6_bench_early_reads.json
python -c "
import json; d=json.loads(open('$O/r6_bench_early_reads.json').read()); c=d['ceilings']; print('default', round(d['value'],1), d['ms_per_step'], (d.get('parity_in_run') or {}).get('ok'), d['roofline']['frac'], d.get('kernel_family_ms'), 'mfma sust', c['mfma_sustained_TFLOPs'], 'sclk', c['sclk_MHz_mean'])"
