#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "gemv or few_rows or small_batch or mid_batch_streamer or skinny or ln_linear or linear2 or ln_qkv" ) > $O/r2_pytest6.log 2>&1
echo "pytest rc=$?" >> $O/r2_pytest6.log; tail -4 $O/r2_pytest6.log
for B in 1 2 4 8 16; do
  timeout 200 python bench.py --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r2_small_b$B.json 2> $O/r2_small_b$B.err
  python -c "import json;d=json.loads(open('$O/r2_small_b$B.json').read().strip().splitlines()[-1]);print('B',$B,'tok/s',round(d['value'],1),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1))"
done
# harness counterparts on the real config (BASELINE configs[1] and [2]); the auto-tune fills the HBM like the reference's
timeout 1500 python -m block_transformer_amd.measure_generation_time --config-name block_main_b4_1.2b ++benchmark_prefill_length=2048 ++benchmark_decode_length=128 ++save_dir=$O/r2_results ++output_dir=p2048_d128 > $O/r2_measure_p2048_d128.log 2>&1
echo "measure p2048/d128 rc=$?"; tail -5 $O/r2_measure_p2048_d128.log; cat $O/r2_results/p2048_d128/generation_time.csv
