#!/bin/bash
# round 3, GPU call 15: batch sweep on the new dispatch (regression check vs the round-2 table) + other BASELINE configurations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for B in 1 8 64 256 1024; do
  timeout 300 python bench.py --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r3_sweep_b$B.json 2> $O/r3_sweep_b$B.err
  python -c "import json;d=json.loads(open('$O/r3_sweep_b$B.json').read().strip().splitlines()[-1]);print('B',$B,'tok/s',round(d['value']),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1),'decode tok/s',round(d['decode_phase_tokens_per_s_per_gpu']))"
done
timeout 400 python bench.py --prefill 128 --decode 2048 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > $O/r3_bench_p128_d2048.json 2> $O/r3_bench_p128_d2048.err
python -c "import json;d=json.loads(open('$O/r3_bench_p128_d2048.json').read().strip().splitlines()[-1]);print('P128/D2048 B',d['config']['batch_per_gpu'],'tok/s',round(d['value']),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1))"
