#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "mid_rows or mid_batch or ring_depth or batch_sizes_around or linear_splitk or gemv or streamer_stack or measure_generation" ) > $O/r2_pytest7.log 2>&1
echo "pytest rc=$?" >> $O/r2_pytest7.log; tail -6 $O/r2_pytest7.log
for B in 8 16 32 64 128 256; do
  timeout 200 python bench.py --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r2_mid2_b$B.json 2> $O/r2_mid2_b$B.err
  python -c "import json;d=json.loads(open('$O/r2_mid2_b$B.json').read().strip().splitlines()[-1]);print('B',$B,'tok/s',round(d['value']),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1),'decode tok/s',round(d['decode_phase_tokens_per_s_per_gpu']))"
done
timeout 1500 python -m block_transformer_amd.measure_generation_time --config-name block_main_b4_1.2b ++benchmark_prefill_length=2048 ++benchmark_decode_length=128 ++save_dir=$O/r2_results ++output_dir=p2048_d128_v2 > $O/r2_measure_p2048_d128_v2.log 2>&1
echo "measure p2048/d128 rc=$?"; grep -i "memory\|End of search\|Total time" $O/r2_measure_p2048_d128_v2.log | tail -6; cat $O/r2_results/p2048_d128_v2/generation_time.csv
