#!/bin/bash
# round 3, GPU call 20: counters on the VENDOR GEMM (torch F.linear -> hipBLASLt) at two benchmark shapes, same sets as r3_pmc.sh:
# does it win by duty cycle or by clock?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
declare -A SHAPE=( [plain]="--m 32768 --n 6144 --k 2048" [k8192]="--m 32768 --n 2048 --k 8192" [dec]="--m 4096 --n 6144 --k 2048" )
for name in plain k8192 dec; do
  for set in A B F; do
    case $set in A) C="$SQA";; B) C="$SQB";; F) C="FETCH_SIZE";; esac
    ( cd /tmp && timeout 150 rocprofv3 --pmc $C -d "$GRAFT_REPO_ROOT/$O/r3_pmc_vendor_${name}_$set" -- python "$GRAFT_REPO_ROOT/tools/vendor_pmc_target.py" ${SHAPE[$name]} ) > $O/r3_pmc_vendor_${name}_$set.log 2>&1
    python tools/summarize_prof.py pmc $O/r3_pmc_vendor_${name}_$set/*/ --kernel "Cijk" > $O/r3_pmc_vendor_${name}_$set.json 2>&1
  done
done
rm -rf $O/r3_pmc_vendor_*/
for f in $O/r3_pmc_vendor_*_A.json; do echo $f; cat $f | tr -d '\n ' | cut -c1-900; echo; done
for f in $O/r3_pmc_vendor_*_B.json $O/r3_pmc_vendor_*_F.json; do echo $f; cat $f | tr -d '\n ' | cut -c1-600; echo; done
