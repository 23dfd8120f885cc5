#!/bin/bash
# round 3, GPU call 10: bt_logits_argmax: per-op + model tests, A/B of the bench with and without materialised logits
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x \
   -k "logits_argmax or keeping_logits or argmax or benchmark_batch or g6_g9 or token_decoder_generate" ) > $O/r3_pytest10.log 2>&1
echo "pytest rc=$?" >> $O/r3_pytest10.log
tail -5 $O/r3_pytest10.log
for KL in "" "--keep-logits" "" "--keep-logits"; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $KL > $O/r3_bench_kl.json 2> $O/r3_bench_kl.err
  python -c "
import json
d=json.loads(open('gpurun_out/r3_bench_kl.json').read().strip().splitlines()[-1])
print('keep_logits', d['config']['logits_materialised'], 'tok/s',round(d['value']),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1), {k: d['kernel_family_ms'][k] for k in ('linear_mfma','argmax')})"
done
