#!/bin/bash
# round 6, GPU call 10: the sticky word in its own cache line + agent-scope read: the P128/D2048 command that raised a false time-out 4 times out of 5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for i in 1 2; do
BT_SK_DIAG=1 timeout 600 python tools/bench_knobs.py serial_epilogues=0 -- --prefill 128 --decode 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings > $O/r6_p128_dbg_$i.json 2> $O/r6_p128_dbg_$i.err
echo "dbg run $i rc=$?"; grep "bench_knobs\|timed steps\|Error" $O/r6_p128_dbg_$i.err | tail -5
timeout 600 python bench.py --prefill 128 --decode 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-ceilings > $O/r6_p128_product_$i.json 2> $O/r6_p128_product_$i.err
echo "product run $i rc=$?"; grep "timed steps\|Error\|parity" $O/r6_p128_product_$i.err | tail -4
done
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "stream_k or handoff" 2>&1 | tail -3
