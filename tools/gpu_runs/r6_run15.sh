#!/bin/bash
# round 6, GPU call 15: the mid-batch path with its tile counters cleared by the library kernel instead of a memset node
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "mid_ or batch_sizes_around or every_reference or few_rows or unwritten or small_batch" 2>&1 | tail -3
for B in 8 64 256; do
  timeout 200 python bench.py --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-ceilings 2>/dev/null | tail -1 > $O/r6_sweep2_b$B.json
  python -c "
import json; d=json.loads(open('$O/r6_sweep2_b$B.json').read()); print($B, round(d['value'],1), (d.get('parity_in_run') or {}).get('ok'))"
done
