#!/bin/bash
# round 6, GPU call 31: PMC passes over the fused mid-batch launch at 256 rows, launched eagerly (tools/mid_launch_probe.py; separate --pmc runs, no trace domains)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 200 python tools/mid_launch_probe.py --rows 256 --iters 3 > $O/r6_mid_probe_plain.log 2>&1; echo "plain rc=$?"; tail -1 $O/r6_mid_probe_plain.log
SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for set in F T L A X; do
  case $set in F) C="FETCH_SIZE";; T) C="TCC_HIT_sum TCC_MISS_sum";; L) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS";; A) C="$SQA";; X) C="TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum";; esac
  rm -rf $O/r6_pmc_midp_$set
  ( cd /tmp && timeout 200 rocprofv3 --pmc $C -d "$GRAFT_REPO_ROOT/$O/r6_pmc_midp_$set" -- python "$GRAFT_REPO_ROOT/tools/mid_launch_probe.py" --rows 256 --iters 10 ) > $O/r6_pmc_midp_$set.log 2>&1
  echo "set $set rc=$?"
  python tools/summarize_prof.py pmc $O/r6_pmc_midp_$set/*/ --kernel "gemm_bf16_mfma256x128_slab_kernel<true, true>" > $O/r6_pmc_mid_fused_$set.json 2>&1
  python tools/summarize_prof.py pmc $O/r6_pmc_midp_$set/*/ --kernel "gemm_bf16_mfma256x128_slab_kernel<true, false>" > $O/r6_pmc_mid_res_$set.json 2>&1
  rm -rf $O/r6_pmc_midp_$set
  cat $O/r6_pmc_mid_fused_$set.json | tr -d '\n ' | cut -c1-500; echo
  cat $O/r6_pmc_mid_res_$set.json | tr -d '\n ' | cut -c1-500; echo
done
