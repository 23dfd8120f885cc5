#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "linear" ) 2>&1 | tail -4
( timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x -k "large_stack or real_dims or full_depth or production_dims or long_decode" ) 2>&1 | tail -4
timeout 300 python tools/kernel_probe.py lin2 --iters 10 > $O/r2_probe_lin2.jsonl 2> $O/r2_probe_lin2.err; cat $O/r2_probe_lin2.jsonl; tail -3 $O/r2_probe_lin2.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r2_bench_lin2.json 2> $O/r2_bench_lin2.err; cat $O/r2_bench_lin2.json; tail -3 $O/r2_bench_lin2.err
