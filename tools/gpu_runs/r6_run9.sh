#!/bin/bash
# round 6, GPU call 9: which pass sees a non-zero sticky word, and what does it hold?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
BT_SK_DIAG=1 timeout 600 python tools/bench_knobs.py serial_epilogues=0 -- --prefill 128 --decode 2048 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings > $O/r6_p128_dbg.json 2> $O/r6_p128_dbg.err
grep "bench_knobs\|warmup" $O/r6_p128_dbg.err | tail -14
