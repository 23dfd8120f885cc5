#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "attention or token_decoder or g6_g9 or bf16 or eos or full_depth or real_dims" ) > $O/r2_pytest13.log 2>&1
echo "pytest rc=$?" >> $O/r2_pytest13.log; tail -5 $O/r2_pytest13.log
python tools/kernel_probe.py local --batch 4096 --iters 20 --product > $O/r2_probe_local_new.jsonl 2>/dev/null; cat $O/r2_probe_local_new.jsonl
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r2_bench_localk.json 2> $O/r2_bench_localk.err
python -c "import json;d=json.loads(open('$O/r2_bench_localk.json').read().strip().splitlines()[-1]);print(round(d['value']),round(d['prefill_ms'],1),round(d['decode_ms'],1),d['kernel_family_ms'])"
for B in 1 64; do
  timeout 200 python bench.py --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r2_localk_b$B.json 2> $O/r2_localk_b$B.err
  python -c "import json;d=json.loads(open('$O/r2_localk_b$B.json').read().strip().splitlines()[-1]);print('B',$B,'tok/s',round(d['value'],1),'decode',round(d['decode_ms'],1))"
done
