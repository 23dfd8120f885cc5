#!/bin/bash
# round 6, GPU call 33: stream form with A eight K-tiles ahead too: oracle tests at the 1.2B width, kernel stats at B = 256, A/B vs the LDS-ring half mode at B = 256
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -f $O/r6_ab_mid_stream8_*.json
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "1p2b_width" > $O/r6_run33_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/r6_run33_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
B=256
rm -rf $O/r6_prof_b$B
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r6_prof_b$B" -- python "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r6_prof_b$B.log 2>&1
python tools/summarize_prof.py stats $O/r6_prof_b$B/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --batch $B (P2048/D128), round 6, stream form with A and W eight K-tiles ahead" --top 16 > $O/r6_bench_b${B}_kernel_stats_stream8.md 2>&1
rm -rf $O/r6_prof_b$B
grep "slab_kernel\|mid_stream" $O/r6_bench_b256_kernel_stats_stream8.md | head -3 | cut -c1-200
for v in 4 1 4 1; do
  timeout 200 python tools/bench_knobs.py mid_tall_fuse=$v -- --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | tail -1 > $O/r6_ab_mid_stream8_${v}_b${B}_$RANDOM.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6_ab_mid_stream8_*_b*.json')):
    d = json.loads(open(f).read())
    print(f.split('/')[-1], round(d['value'], 1), 'decode-phase', round(d.get('decode_phase_tokens_per_s_per_gpu', 0), 1))
PY
