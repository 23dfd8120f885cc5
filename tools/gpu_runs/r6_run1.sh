#!/bin/bash
# round 6, GPU call 1: the hand-off time-out path (kernel + engine + module forward), the stream-K / mid-batch tests around the ADVICE r5
# edits (split-K room, fp32 workspace, 257..512-row precondition), and this box's baseline line of the round-5 kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "stream_k or linear" > $O/r6_run1_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -4 $O/r6_run1_kernel_tests.log
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "handoff or device_guard or large_stack or mid_rows or batch_sizes_around or scoring_path or every_reference_config" > $O/r6_run1_model_tests.log 2>&1
echo "model tests rc=$?"; tail -4 $O/r6_run1_model_tests.log
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/r6_bench_baseline_r5_kernels.json
python -c "
import json; d=json.loads(open('$O/r6_bench_baseline_r5_kernels.json').read()); print('default', round(d['value'],1), d['ms_per_step'], (d.get('parity_in_run') or {}).get('ok'), d['roofline']['frac'], d.get('kernel_family_ms'))"
