#!/bin/bash
# round 3, GPU call 9: full GPU suite + smoke + default bench on the tree with the second-generation persistent GEMM
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/r3_pytest_full1.log 2>&1
echo "pytest rc=$?" >> $O/r3_pytest_full1.log
tail -6 $O/r3_pytest_full1.log
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py > $O/r3_bench_default1.json 2> $O/r3_bench_default1.err
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench_default1.json').read().strip().splitlines()[-1])
print('tok/s',round(d['value']),'ms/step',round(d['ms_per_step'],1),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1))
print(d['kernel_family_ms']); print('frac', d['roofline']['frac'], 'per-rank', d['per_rank_tokens_per_s'])"
tail -3 $O/r3_bench_default1.err
