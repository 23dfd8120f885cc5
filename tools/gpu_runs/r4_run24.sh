#!/bin/bash
# round 4, GPU call 24 (final tree): bench.py at B = 1 / 2 / 4 for the sweep table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for b in 1 2 4; do
  timeout 60 python bench.py --batch $b --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --no-parity --no-secondary > $O/r4_sweep_final_b$b.json 2> $O/r4_sweep_final_b$b.log
  python -c "import json;d=json.load(open('$O/r4_sweep_final_b$b.json'));print('B=$b', round(d['value'],1), 'tok/s', round(d['ms_per_step'],2), 'ms')"
done
