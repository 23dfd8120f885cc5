#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/r2_pytest8_full.log 2>&1
echo "pytest rc=$?" >> $O/r2_pytest8_full.log; tail -8 $O/r2_pytest8_full.log
for B in 32 64 128 256; do
  timeout 200 python bench.py --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r2_mid3_b$B.json 2> $O/r2_mid3_b$B.err
  python -c "import json;d=json.loads(open('$O/r2_mid3_b$B.json').read().strip().splitlines()[-1]);print('B',$B,'tok/s',round(d['value']),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1),'decode tok/s',round(d['decode_phase_tokens_per_s_per_gpu']))"
done
timeout 900 python -m block_transformer_amd.measure_generation_time --config-name block_main_b4_1.2b ++benchmark_prefill_length=128 ++benchmark_decode_length=2048 ++save_dir=$O/r2_results ++output_dir=p128_d2048 "++benchmark_batch_sizes=[1,64,1024,4096]" > $O/r2_measure_p128_d2048.log 2>&1
echo "measure p128/d2048 rc=$?"; cat $O/r2_results/p128_d2048/generation_time.csv
timeout 300 python -m block_transformer_amd.inference_demo --model block_main_b4_1.2b --random-init --batch_size 128 --max_length 1334 --no-stream > $O/r2_inference_demo_b128.log 2>&1
echo "demo rc=$?"; tail -8 $O/r2_inference_demo_b128.log
timeout 120 python -m block_transformer_amd.inference_demo --model block_main_b4_1.2b --random-init --batch_size 4 --max_length 96 --prompt-length 32 > $O/r2_inference_demo_b4_stream.log 2>&1
echo "demo stream rc=$?"; tail -c 600 $O/r2_inference_demo_b4_stream.log
