#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "few_rows or small_batch or gemv or ln_qkv or ln_linear or g6_g9 or token_decoder_generate or in_launch" ) > $O/r2_pytest9.log 2>&1
echo "pytest rc=$?" >> $O/r2_pytest9.log; tail -5 $O/r2_pytest9.log
for B in 1 2 4; do
  timeout 200 python bench.py --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r2_small2_b$B.json 2> $O/r2_small2_b$B.err
  python -c "import json;d=json.loads(open('$O/r2_small2_b$B.json').read().strip().splitlines()[-1]);print('B',$B,'tok/s',round(d['value'],1),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1))"
done
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r2_prof_b1" -- python "$GRAFT_REPO_ROOT/bench.py" --batch 1 --steps 1 --warmup 1 --no-roofline --no-cpu-baseline ) > $O/r2_prof_b1.log 2>&1
python tools/summarize_prof.py stats $O/r2_prof_b1/*/ --title "bench.py --batch 1 (P2048/D128), 3 generate() calls, round 2" --top 14 > $O/r2_bench_b1_kernel_stats.md 2>&1
head -24 $O/r2_bench_b1_kernel_stats.md
