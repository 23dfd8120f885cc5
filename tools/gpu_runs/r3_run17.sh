#!/bin/bash
# round 3, GPU call 17: counter evidence on the final kernels (profiles/r3_pmc_*.json) + vendor yardstick on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
GIT=${1:-unknown}
time bash tools/gpu_runs/r3_pmc.sh $GIT 2>&1 | tail -40
timeout 300 python tools/kernel_probe.py vendor > gpurun_out/r3_vendor_final.jsonl 2> gpurun_out/r3_vendor_final.err
cut -c1-220 gpurun_out/r3_vendor_final.jsonl
du -sh gpurun_out
