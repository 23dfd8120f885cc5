#!/bin/bash
# round 3, GPU call 23: bench line with the fast K-tiles
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline > $O/r3_bench_fast_ktiles.json 2> $O/r3_bench_fast_ktiles.err
cut -c1-600 $O/r3_bench_fast_ktiles.json; tail -4 $O/r3_bench_fast_ktiles.err
