#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "layernorm or flash" ) 2>&1 | tail -4
timeout 200 python tools/kernel_probe.py ln --iters 20 > $O/r2_probe_ln.jsonl 2> $O/r2_probe_ln.err; cat $O/r2_probe_ln.jsonl; tail -3 $O/r2_probe_ln.err
timeout 300 python tools/kernel_probe.py flash --batch 256 --keys 512 --iters 10 > $O/r2_probe_flash_opt.jsonl 2> $O/r2_probe_flash_opt.err; cat $O/r2_probe_flash_opt.jsonl; tail -3 $O/r2_probe_flash_opt.err
