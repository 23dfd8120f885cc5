#!/bin/bash
# round 4, GPU call 10: two-stream parallel-residual branch: bit-identity, then the headline bench A/B (on / off / on / off, same box)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "two_branch" 2>&1 | tail -12 > $O/r4_t10.log
cat $O/r4_t10.log
for rep in 1 2; do
for tb in on off; do
  timeout 400 python bench.py --steps 3 --warmup 1 --two-branch $tb --no-cpu-baseline --no-parity --no-secondary > $O/r4_bench_2br_${tb}_$rep.json 2> $O/r4_bench_2br_${tb}_$rep.log
  python -c "import json;d=json.load(open('$O/r4_bench_2br_${tb}_$rep.json'));print('two_branch $tb', round(d['value'],1), 'tok/s', 'prefill', round(d['prefill_ms'],1), 'decode', round(d['decode_ms'],1), d['config'].get('two_branch'))" || tail -5 $O/r4_bench_2br_${tb}_$rep.log
done
done
