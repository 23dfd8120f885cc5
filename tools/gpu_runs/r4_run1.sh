#!/bin/bash
# round 4, GPU call 1: the new C-ABI compaction / guard / steady-state parity tests, then the default bench with parity_in_run + secondary
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "compact" -x 2>&1 | tail -5 > $O/r4_t1_kernels.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -x -s -k "compaction or device_guard or steady_state or streamer or sampling or eos_handling or shard_equivalence or demo_main" 2>&1 | grep -v "^$" | tail -40 > $O/r4_t1_model.log
timeout 900 python bench.py --steps 3 --warmup 1 > $O/r4_bench_first.json 2> $O/r4_bench_first.log
echo "bench rc $?" >> $O/r4_bench_first.log
cat $O/r4_t1_kernels.log; cat $O/r4_t1_model.log; tail -25 $O/r4_bench_first.log; head -c 1500 $O/r4_bench_first.json
