#!/bin/bash
# round 3, GPU call 25 (final tree): full -m gpu suite, smoke, default bench, rocprofv3 kernel stats, PMC passes, vendor yardstick
cd "$GRAFT_REPO_ROOT" || exit 1
GIT=${1:-unknown}
bash tools/gpu_runs/r3_run18.sh
bash tools/gpu_runs/r3_pmc.sh $GIT 2>&1 | tail -12
timeout 300 python tools/kernel_probe.py vendor > gpurun_out/r3_vendor_final.jsonl 2> gpurun_out/r3_vendor_final.err
cut -c1-220 gpurun_out/r3_vendor_final.jsonl
