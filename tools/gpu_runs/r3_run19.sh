#!/bin/bash
# round 3, GPU call 19 (final tree): the other BASELINE configurations + the batch sweep + keep-logits A/B + the harness with profiling
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2 B',d['config']['batch_per_gpu'],'tok/s',round(d['value']),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1),'decode tok/s',round(d['decode_phase_tokens_per_s_per_gpu']))"; }
timeout 400 python bench.py --prefill 128 --decode 2048 --steps 2 --warmup 1 --no-cpu-baseline > $O/r3_bench_p128_d2048_final.json 2> $O/r3_bench_p128_d2048_final.err; show $O/r3_bench_p128_d2048_final.json P128/D2048
timeout 500 python bench.py --prefill 2048 --decode 2048 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > $O/r3_bench_p2048_d2048_final.json 2> $O/r3_bench_p2048_d2048_final.err; show $O/r3_bench_p2048_d2048_final.json P2048/D2048
timeout 300 python bench.py --config block_main_b8_1.2b --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r3_bench_b8_final.json 2> $O/r3_bench_b8_final.err; show $O/r3_bench_b8_final.json L8
timeout 300 python bench.py --keep-logits --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r3_bench_keep_logits.json 2> $O/r3_bench_keep_logits.err; show $O/r3_bench_keep_logits.json keep-logits
for B in 1 2 4 8 16 32 64 128 256 1024; do
  timeout 300 python bench.py --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r3_sweep_final_b$B.json 2> $O/r3_sweep_final_b$B.err
  show $O/r3_sweep_final_b$B.json sweep
done
mkdir -p $O/r3_results
timeout 600 python -m block_transformer_amd.measure_generation_time --config-name block_main_b4_1.2b "++benchmark_batch_sizes=[64,4096]" ++profiling=true ++log_path=$O/r3_results/traces ++save_dir=$O/r3_results > $O/r3_results/generation_time_profiled.log 2>&1
tail -30 $O/r3_results/generation_time_profiled.log
ls -la $O/r3_results/traces; gzip -9 $O/r3_results/traces/*.log; du -sh $O/r3_results
