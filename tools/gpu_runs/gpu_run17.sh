#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "linear or gemv or few_rows or layernorm" ) 2>&1 | tail -4
( timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x -k "small_batch or dual or golden or g6 or G6" ) 2>&1 | tail -4
for B in 1 4; do
  timeout 200 python bench.py --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r2_bench_gemvpipe_b$B.json 2> $O/r2_bench_gemvpipe_b$B.err
  python -c "import json;d=json.loads(open('$O/r2_bench_gemvpipe_b$B.json').read().strip().splitlines()[-1]);print('B',$B,round(d['value'],1),round(d['prefill_ms'],1),round(d['decode_ms'],1))"
done
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r2_bench_ln_reg.json 2> $O/r2_bench_ln_reg.err
python -c "import json;d=json.loads(open('$O/r2_bench_ln_reg.json').read().strip().splitlines()[-1]);print(round(d['value'],1),round(d['prefill_ms'],1),round(d['decode_ms'],1),d['kernel_family_ms'])"
