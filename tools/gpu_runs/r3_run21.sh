#!/bin/bash
# round 3, GPU call 21: fast (static ring stage, branch-free) steady-state K-tiles of the persistent GEMM: A/B vs the generic loop
# (variant 18), bit-identity, fp64 kernel tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 400 python tools/kernel_probe.py ppab --variant 12 --ab-other 18 > $O/r3_ppab_fast.jsonl 2> $O/r3_ppab_fast.err
python -c "
import json
for l in open('gpurun_out/r3_ppab_fast.jsonl'):
    r = json.loads(l); print(r['M'], r['N'], r['K'], r['mode'], 'fast', min(r['a_us']), 'generic', min(r['b_us']), 'x%.3f' % (min(r['b_us'])/min(r['a_us'])), 'TF', r['a_TF'], 'same' if r['bit_identical'] else 'DIFFERENT')"
tail -3 $O/r3_ppab_fast.err
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "linear or qkv_rope_append or logits_argmax or gelu" ) > $O/r3_pytest21.log 2>&1
tail -5 $O/r3_pytest21.log
mkdir -p $O/r3_results
timeout 600 python -m block_transformer_amd.measure_generation_time --config-name block_main_b4_1.2b "++benchmark_batch_sizes=[64,4096]" ++benchmark_prefill_length=2048 ++benchmark_decode_length=128 ++profiling=true ++log_path=$O/r3_results/traces ++save_dir=$O/r3_results > $O/r3_results/generation_time_profiled.log 2>&1
tail -32 $O/r3_results/generation_time_profiled.log
ls -la $O/r3_results/traces; gzip -9 $O/r3_results/traces/*.log; du -sh $O/r3_results
