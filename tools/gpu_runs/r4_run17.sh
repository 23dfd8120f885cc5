#!/bin/bash
# round 4, GPU call 17: mid-batch tests after forcing two K slices up to 224 rows; B = 96 / 128 / 192 bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "mid_batch or mid_rows or unwritten_cache or batch_sizes_around or every_reference_config or compaction" 2>&1 | tail -4
for b in 96 128 192; do
  timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-parity --no-secondary > $O/r4_sweep_b$b.json 2> $O/r4_sweep_b$b.log
  python -c "import json;d=json.load(open('$O/r4_sweep_b$b.json'));print('B=$b', round(d['value'],1), 'tok/s', 'decode', round(d.get('decode_phase_tokens_per_s_per_gpu',0)))"
done
