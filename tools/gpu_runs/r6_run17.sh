#!/bin/bash
# round 6, GPU call 17: the 97..512-row QKV + h->4h launch with fused epilogues (hand-off between a tile's two K ranges): parity of the
# mid-batch paths, then end-to-end A/B one setting per process (bt_debug_mid_tall_fuse)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "mid_ or batch_sizes_around or every_reference or unwritten" > $O/r6_run17_tests.log 2>&1
echo "tests rc=$?"; tail -15 $O/r6_run17_tests.log | cut -c1-300
for B in 128 256 512; do
  for v in 0 1 0 1; do
    timeout 200 python tools/bench_knobs.py mid_tall_fuse=$v -- --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | tail -1 > $O/r6_ab_mid_fuse${v}_b${B}_$RANDOM.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6_ab_mid_fuse*_b*.json')):
    try:
        d = json.loads(open(f).read())
        print(f.split('/')[-1], round(d['value'], 1), 'decode-phase', round(d.get('decode_phase_tokens_per_s_per_gpu', 0), 1))
    except Exception as e:
        print(f, 'unreadable', e)
PY
timeout 200 python bench.py --batch 256 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-ceilings 2>/dev/null | tail -1 > $O/r6_mid_fused_b256_parity.json
python -c "
import json; d=json.loads(open('$O/r6_mid_fused_b256_parity.json').read()); print('B=256 product', round(d['value'],1), (d.get('parity_in_run') or {}))" | cut -c1-400
