#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "few_rows or small_batch or mid_batch_streamer or skinny or ln_qkv or ln_linear or linear2 or batch_sizes_around" ) > $O/r2_pytest10.log 2>&1
echo "pytest rc=$?" >> $O/r2_pytest10.log; tail -5 $O/r2_pytest10.log
for B in 5 8 12 16; do
  timeout 200 python bench.py --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r2_small3_b$B.json 2> $O/r2_small3_b$B.err
  python -c "import json;d=json.loads(open('$O/r2_small3_b$B.json').read().strip().splitlines()[-1]);print('B',$B,'tok/s',round(d['value'],1),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1))"
done
