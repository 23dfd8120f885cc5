#!/bin/bash
# round-2 GPU run 1: full GPU suite on the refactored engine, bench with / without the half-batch pipeline, vendor kernel names
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/r2_pytest1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest1.log
tail -5 gpurun_out/r2_pytest1.log
for mode in off on; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipeline $mode > gpurun_out/r2_bench_pipe_$mode.json 2> gpurun_out/r2_bench_pipe_$mode.err
  echo "bench pipeline=$mode rc=$?"; tail -c 600 gpurun_out/r2_bench_pipe_$mode.json
done
BT_PIPELINE_SYNC=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --pipeline on > gpurun_out/r2_bench_pipe_sync.json 2> gpurun_out/r2_bench_pipe_sync.err
echo "bench pipeline=sync rc=$?"; tail -c 400 gpurun_out/r2_bench_pipe_sync.json
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2_vendor_trace" -- python "$GRAFT_REPO_ROOT/tools/vendor_trace.py" ) > gpurun_out/r2_vendor_trace.log 2>&1
tail -12 gpurun_out/r2_vendor_trace.log
