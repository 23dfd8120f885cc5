#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x -k "full_depth or real_dims or production_dims or golden or G2 or g2 or prefill or block_decoder" ) 2>&1 | tail -4
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r2_bench_flash95.json 2> $O/r2_bench_flash95.err
python -c "import json;d=json.loads(open('$O/r2_bench_flash95.json').read().strip().splitlines()[-1]);print(round(d['value'],1),round(d['prefill_ms'],1),round(d['decode_ms'],1),d['kernel_family_ms'])"
