#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "flash_prefill and flash_v3" ) 2>&1 | tail -3
for cfg in "256 512" "64 512" "128 1024"; do
  set -- $cfg
  timeout 100 python tools/kernel_probe.py flash --batch $1 --keys $2 --iters 10 2>/dev/null | tail -3
done > $O/r2_probe_flash_v3d.jsonl
cat $O/r2_probe_flash_v3d.jsonl
( cd /tmp && timeout 150 rocprofv3 --pmc FETCH_SIZE -d "$GRAFT_REPO_ROOT/$O/r2_pmc_flash_F2" -- python "$GRAFT_REPO_ROOT/tools/kernel_probe.py" flash --batch 256 --keys 512 --iters 3 ) > $O/r2_pmc_flash_F2.log 2>&1
python tools/summarize_prof.py pmc $O/r2_pmc_flash_F2/*/ --kernel "flash_prefill_v3_kernel<128" 2>&1 | head -12
rm -rf $O/r2_pmc_flash_F2
