#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 150 rocprofv3 --pmc $C -d "$GRAFT_REPO_ROOT/$O/r2_pmc_flash_$C" -- python "$GRAFT_REPO_ROOT/tools/kernel_probe.py" flash --batch 256 --keys 512 --iters 3 ) > $O/r2_pmc_flash_$C.log 2>&1
  for K in "flash_prefill_v2_kernel<128, 4, 2, 31" "flash_prefill_v2_kernel<128, 4, 2, 95" "flash_prefill_v3_kernel<128"; do
    echo "== $C :: $K"
    python tools/summarize_prof.py pmc $O/r2_pmc_flash_$C/*/ --kernel "$K" 2>&1 | head -20
  done
  rm -rf $O/r2_pmc_flash_$C
done > $O/r2_pmc_flash.txt 2>&1
cat $O/r2_pmc_flash.txt
