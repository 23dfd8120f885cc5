#!/bin/bash
# round 3, GPU call 24: s_setprio A/B on the one-tile 256 x 128 kernel; production (no setprio) vs round-2 setting on the persistent kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 200 python tools/kernel_probe.py t21 > $O/r3_t21_setprio.jsonl 2> $O/r3_t21_setprio.err; cut -c1-400 $O/r3_t21_setprio.jsonl; tail -2 $O/r3_t21_setprio.err
bash tools/gpu_runs/r3_run22.sh 24
