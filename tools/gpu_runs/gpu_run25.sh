#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "flash" ) 2>&1 | tail -3
( timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x -k "ragged or mask or pad or golden or G2 or g2 or eos or full_depth or real_dims" ) 2>&1 | tail -3
for cfg in "256 512" "64 512" "1024 128"; do
  set -- $cfg
  timeout 100 python tools/kernel_probe.py flash --batch $1 --keys $2 --iters 10 2>/dev/null | tail -2
done > $O/r2_probe_flash_v3c.jsonl
cat $O/r2_probe_flash_v3c.jsonl
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r2_bench_flash_v3.json 2> $O/r2_bench_flash_v3.err
python -c "import json;d=json.loads(open('$O/r2_bench_flash_v3.json').read().strip().splitlines()[-1]);print(round(d['value'],1),round(d['prefill_ms'],1),round(d['decode_ms'],1),d['kernel_family_ms'])"
