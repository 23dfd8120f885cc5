#!/bin/bash
# round 6, GPU call 32: stream form of the fused mid-batch launch at 129..256 rows (W global -> registers): oracle tests at the 1.2B width, A/B vs the
# LDS-ring half mode (fuse 4) and halves off (3) at B = 160 / 192 / 256, kernel stats at B = 256
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -f $O/r6_ab_mid_stream_*.json
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "1p2b_width or mid_rows_fused" > $O/r6_run32_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $O/r6_run32_tests.log | cut -c1-300
[ $rc -ne 0 ] && { grep -n "Error\|assert" $O/r6_run32_tests.log | head -20; exit 1; }
for B in 160 192 256; do
  for v in 4 1 3 4 1 3; do
    timeout 200 python tools/bench_knobs.py mid_tall_fuse=$v -- --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | tail -1 > $O/r6_ab_mid_stream_${v}_b${B}_$RANDOM.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6_ab_mid_stream_*_b*.json')):
    try:
        d = json.loads(open(f).read())
        print(f.split('/')[-1], round(d['value'], 1), 'decode-phase', round(d.get('decode_phase_tokens_per_s_per_gpu', 0), 1))
    except Exception as e:
        print(f, 'unreadable', e)
PY
for B in 192 256; do
timeout 200 python bench.py --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-ceilings 2>/dev/null | tail -1 > $O/r6_sweep_stream_b$B.json
python -c "
import json; d=json.loads(open('$O/r6_sweep_stream_b$B.json').read()); p=d.get('parity_in_run') or {}; print('B=$B product', round(d['value'],1), p.get('ok'), p.get('ctx_err'), p.get('logit_err'))"
done
B=256
rm -rf $O/r6_prof_b$B
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r6_prof_b$B" -- python "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r6_prof_b$B.log 2>&1
python tools/summarize_prof.py stats $O/r6_prof_b$B/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings (P2048/D128), round 6, stream form of the fused mid-batch launch" --top 16 > $O/r6_bench_b${B}_kernel_stats_stream.md 2>&1
rm -rf $O/r6_prof_b$B
grep "slab_kernel\|mid_stream" $O/r6_bench_b256_kernel_stats_stream.md | cut -c1-200
