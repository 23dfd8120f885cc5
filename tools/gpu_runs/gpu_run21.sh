#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
for B in 64 256; do
  timeout 200 python bench.py --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r2_bench_mid2_b$B.json 2> $O/r2_bench_mid2_b$B.err
  python -c "import json;d=json.loads(open('$O/r2_bench_mid2_b$B.json').read().strip().splitlines()[-1]);print('B',$B,round(d['value'],1),round(d['prefill_ms'],1),round(d['decode_ms'],1),round(d['decode_phase_tokens_per_s_per_gpu']))"
done
