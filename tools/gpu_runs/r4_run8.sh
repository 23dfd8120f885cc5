#!/bin/bash
# round 4, GPU call 8: whole model-level GPU suite on the new small- / mid-batch dispatch (cutoff 2), then the small-batch sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -15 > $O/r4_t8_model.log
for b in 1 2 3 4 8 16 32 64 128 256; do
  timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-parity --no-secondary > $O/r4_sweep_b$b.json 2> $O/r4_sweep_b$b.log
done
cat $O/r4_t8_model.log
for b in 1 2 3 4 8 16 32 64 128 256; do python -c "import json;d=json.load(open('$O/r4_sweep_b$b.json'));print('B=$b', round(d['value'],1), 'tok/s', 'decode', round(d.get('decode_phase_tokens_per_s_per_gpu',0)))"; done
