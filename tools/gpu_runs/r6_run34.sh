#!/bin/bash
# round 6, GPU call 34: product back on the LDS-ring half mode, stream form as debug knob 5: the 1.2B-width oracle tests for both, mid-batch tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "1p2b_width or mid_rows_fused or batch_sizes_around" > $O/r6_run34_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/r6_run34_tests.log | cut -c1-300
timeout 200 python bench.py --batch 256 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-ceilings 2>/dev/null | tail -1 > $O/r6_sweep_final2_b256.json
python -c "
import json; d=json.loads(open('$O/r6_sweep_final2_b256.json').read()); p=d.get('parity_in_run') or {}; print('B=256 product', round(d['value'],1), p.get('ok'), p.get('ctx_err'), p.get('logit_err'))"
