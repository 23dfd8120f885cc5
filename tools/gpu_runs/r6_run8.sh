#!/bin/bash
# round 6, GPU call 8: the P128/D2048 bench command that flagged a hand-off time-out in calls 4 / 5, on the current tree: debug library with
# the device-side diag record, then the product library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
BT_SK_DIAG=1 timeout 600 python tools/bench_knobs.py serial_epilogues=0 -- --prefill 128 --decode 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings > $O/r6_p128_dbg.json 2> $O/r6_p128_dbg.err
tail -c 400 $O/r6_p128_dbg.json; echo; grep -v "^\[bench +" $O/r6_p128_dbg.err | tail -12
timeout 600 python bench.py --prefill 128 --decode 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-ceilings > $O/r6_p128_product.json 2> $O/r6_p128_product.err
tail -c 600 $O/r6_p128_product.json; echo; grep -v "^\[bench +" $O/r6_p128_product.err | tail -8
