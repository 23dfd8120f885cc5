#!/bin/bash
# round 4 counter evidence on the kernels that ship (refresh of tools/gpu_runs/r3_pmc.sh on the round-4 tree; the GEMM main loop is unchanged, the local-window kernel now reduces its scores on the DPP path): separate rocprofv3 --pmc passes (no trace domains) around
# tools/kernel_probe.py --product (libbt_hip.so).  Arg 1 = git revision recorded in the json files.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
GIT=${1:-unknown}
O=gpurun_out
mkdir -p $O
SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
pass() {  # tag, counters, probe args...
  local tag=$1; local ctr=$2; shift 2
  ( cd /tmp && timeout 150 rocprofv3 --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/r4_pmc_$tag" -- python "$GRAFT_REPO_ROOT/tools/kernel_probe.py" --product "$@" ) > $O/r4_pmc_$tag.log 2>&1
}
# ---- persistent GEMM, second-generation K-loop: prefill QKV-shape (plain), h->4h (GELU), 4h->h (residual), decode M = 4096 ----
declare -A SHAPE=( [plain]="--m 32768 --n 6144 --k 2048" [gelu]="--m 32768 --n 8192 --k 2048 --act gelu" [res]="--m 32768 --n 2048 --k 8192 --res" [dec]="--m 4096 --n 6144 --k 2048" )
for name in plain dec; do
  for set in A F W; do
    case $set in A) C="$SQA";; B) C="$SQB";; F) C="FETCH_SIZE";; W) C="WRITE_SIZE";; esac
    pass gemm_${name}_$set "$C" gemm ${SHAPE[$name]} --iters 5
    python tools/summarize_prof.py pmc $O/r4_pmc_gemm_${name}_$set/*/ --kernel "gemm_bf16_mfma256pp2" > $O/r4_pmc_gemm_${name}_$set.json 2>&1
  done
done
python tools/summarize_prof.py traffic $O/r4_pmc_gemm_plain_F/*/ $O/r4_pmc_gemm_plain_W/*/ --kernel gemm_bf16_mfma256pp2 \
   --shape "M=32768 N=6144 K=2048 bf16 (prefill QKV shape, plain epilogue)" --algorithmic-bytes 562036736 --key linear_mfma \
   --kernel-name "gemm_bf16_mfma256pp2_kernel<0, 0> (persistent 256x256, second-generation K-loop)" --git $GIT --update $O/r4_pmc_traffic.json > /dev/null
# ---- global attention (decode step), local window, flash prefill ----
for set in A F W; do
  case $set in A) C="$SQA";; F) C="FETCH_SIZE";; W) C="WRITE_SIZE";; esac
  pass attn_$set "$C" attn --batch 1024 --keys 528 --iters 5
  python tools/summarize_prof.py pmc $O/r4_pmc_attn_$set/*/ --kernel "attend_rows_kernel" > $O/r4_pmc_attn_$set.json 2>&1
done
pass local_A "$SQA" local --batch 4096 --iters 5
python tools/summarize_prof.py pmc $O/r4_pmc_local_A/*/ --kernel "local_attend_kernel" > $O/r4_pmc_local_A.json 2>&1
pass local_B "$SQB" local --batch 4096 --iters 5
python tools/summarize_prof.py pmc $O/r4_pmc_local_B/*/ --kernel "local_attend_kernel" > $O/r4_pmc_local_B.json 2>&1
python tools/summarize_prof.py traffic $O/r4_pmc_attn_F/*/ $O/r4_pmc_attn_W/*/ --kernel attend_rows_kernel \
   --shape "B=1024, 16 heads, d=128, n=528 keys, bf16" --algorithmic-bytes 4429185024 --key attn_rows \
   --kernel-name "attend_rows_kernel<bf16, 128, 4, false, false>" --git $GIT --update $O/r4_pmc_traffic.json > /dev/null
# the rocprofv3 databases are large (gpurun merges at most 64 MiB back): keep the json summaries only
rm -rf $O/r4_pmc_*/
cat $O/r4_pmc_traffic.json
for f in $O/r4_pmc_gemm_plain_A.json $O/r4_pmc_gemm_dec_A.json $O/r4_pmc_local_A.json $O/r4_pmc_local_B.json; do echo $f; grep -A8 derived $f | tr -d '\n '; echo; done
tail -3 $O/r4_pmc_gemm_plain_A.log
