#!/bin/bash
# round 6, GPU call 28: the shipped half-mode form again after the experiments (two barriers per K-tile): mid-batch tests, A/B at B = 256, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -f $O/r6_ab_mid_half6_*.json
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "mid_rows_fused or batch_sizes_around" > $O/r6_run28_tests.log 2>&1
rc=$?; echo "mid tests rc=$rc"; tail -3 $O/r6_run28_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
for B in 192 256; do
  for v in 3 1 3 1; do
    timeout 200 python tools/bench_knobs.py mid_tall_fuse=$v -- --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | tail -1 > $O/r6_ab_mid_half6_${v}_b${B}_$RANDOM.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6_ab_mid_half6_*_b*.json')):
    try:
        d = json.loads(open(f).read())
        print(f.split('/')[-1], round(d['value'], 1), 'decode-phase', round(d.get('decode_phase_tokens_per_s_per_gpu', 0), 1))
    except Exception as e:
        print(f, 'unreadable', e)
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r6_gpu_suite.log 2>&1
echo "GPU suite rc=$?"; tail -3 $O/r6_gpu_suite.log
