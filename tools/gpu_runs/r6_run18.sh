#!/bin/bash
# round 6, GPU call 18: fused epilogues kept where a tile has ONE K range (257..512 rows): tests (fuse 0 / 1 / 2 vs oracle), A/B at B = 300 / 384 / 512
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "mid_ or batch_sizes_around or every_reference" > $O/r6_run18_tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/r6_run18_tests.log | cut -c1-300
for B in 300 384 512; do
  for v in 0 1 0 1; do
    timeout 200 python tools/bench_knobs.py mid_tall_fuse=$v -- --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | tail -1 > $O/r6_ab_mid_fuse1k_${v}_b${B}_$RANDOM.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6_ab_mid_fuse1k_*_b*.json')):
    try:
        d = json.loads(open(f).read())
        print(f.split('/')[-1], round(d['value'], 1), 'decode-phase', round(d.get('decode_phase_tokens_per_s_per_gpu', 0), 1))
    except Exception as e:
        print(f, 'unreadable', e)
PY
for B in 256 512; do
timeout 200 python bench.py --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-ceilings 2>/dev/null | tail -1 > $O/r6_sweep_final_b$B.json
python -c "
import json; d=json.loads(open('$O/r6_sweep_final_b$B.json').read()); p=d.get('parity_in_run') or {}; print('B=$B product', round(d['value'],1), p.get('ok'), p.get('ctx_err'), p.get('logit_err'))"
done
