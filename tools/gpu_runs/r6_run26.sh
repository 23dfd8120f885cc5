#!/bin/bash
# round 6, GPU call 26: half mode as a software-pipelined one-barrier loop: tests, timeline, A/B vs halves off at B = 160 / 192 / 256, kernel stats at B = 256
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -f $O/r6_ab_mid_half4_*.json
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "mid_ or batch_sizes_around or every_reference" > $O/r6_run26_tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/r6_run26_tests.log | cut -c1-300
for D in 0 4 16; do
  BT_MID_TL=1 timeout 300 python tools/bench_knobs.py mid_tall_dbg=$D -- --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2> $O/r6_mid_abl2_$D.err | tail -1 > $O/r6_mid_abl2_$D.json
  grep "bench_knobs" $O/r6_mid_abl2_$D.err | sed 's/^\[bench_knobs\] //' > $O/r6_mid_abl2_timeline_$D.json
  python - <<PY
import json
d = json.loads(open('$O/r6_mid_abl2_timeline_$D.json').read())
b = json.loads(open('$O/r6_mid_abl2_$D.json').read())
print('dbg', $D, 'g0', d['group0_median_ticks'], 'decode tok/s', round(b.get('decode_phase_tokens_per_s_per_gpu', 0)))
PY
done
for B in 160 192 256; do
  for v in 3 1 3 1; do
    timeout 200 python tools/bench_knobs.py mid_tall_fuse=$v -- --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | tail -1 > $O/r6_ab_mid_half4_${v}_b${B}_$RANDOM.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6_ab_mid_half4_*_b*.json')):
    try:
        d = json.loads(open(f).read())
        print(f.split('/')[-1], round(d['value'], 1), 'decode-phase', round(d.get('decode_phase_tokens_per_s_per_gpu', 0), 1))
    except Exception as e:
        print(f, 'unreadable', e)
PY
for B in 192 256; do
timeout 200 python bench.py --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-ceilings 2>/dev/null | tail -1 > $O/r6_sweep_half4_b$B.json
python -c "
import json; d=json.loads(open('$O/r6_sweep_half4_b$B.json').read()); p=d.get('parity_in_run') or {}; print('B=$B product', round(d['value'],1), p.get('ok'), p.get('ctx_err'), p.get('logit_err'))"
done
B=256
rm -rf $O/r6_prof_b$B
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r6_prof_b$B" -- python "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r6_prof_b$B.log 2>&1
python tools/summarize_prof.py stats $O/r6_prof_b$B/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings (P2048/D128), round 6, half mode as a one-barrier software-pipelined loop" --top 16 > $O/r6_bench_b${B}_kernel_stats_fused4.md 2>&1
rm -rf $O/r6_prof_b$B
grep "slab_kernel" $O/r6_bench_b256_kernel_stats_fused4.md | cut -c1-200
