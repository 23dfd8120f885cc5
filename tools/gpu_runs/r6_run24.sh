#!/bin/bash
# round 6, GPU call 24: HBM fetch bytes of the fused mid-batch launch at B = 256 (do the two halves of a column tile share W in their L2?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for set in F T; do
  case $set in F) C="FETCH_SIZE";; T) C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum";; esac
  rm -rf $O/r6_pmc_mid_$set
  ( cd /tmp && timeout 500 rocprofv3 --pmc $C -d "$GRAFT_REPO_ROOT/$O/r6_pmc_mid_$set" -- python "$GRAFT_REPO_ROOT/bench.py" --batch 256 --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r6_pmc_mid_$set.log 2>&1
  python tools/summarize_prof.py pmc $O/r6_pmc_mid_$set/*/ --kernel "gemm_bf16_mfma256x128_slab_kernel<true, true>" > $O/r6_pmc_mid_fused_$set.json 2>&1
  python tools/summarize_prof.py pmc $O/r6_pmc_mid_$set/*/ --kernel "gemm_bf16_mfma256x128_slab_kernel<true, false>" > $O/r6_pmc_mid_res_$set.json 2>&1
  rm -rf $O/r6_pmc_mid_$set
  cat $O/r6_pmc_mid_fused_$set.json | tr -d '\n ' | cut -c1-400; echo
  cat $O/r6_pmc_mid_res_$set.json | tr -d '\n ' | cut -c1-400; echo
done
tail -3 $O/r6_pmc_mid_T.log | cut -c1-300
