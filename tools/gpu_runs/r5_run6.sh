#!/bin/bash
# round 5, GPU call 8: the whole GPU test suite on the round-5 tree, rocprofv3 kernel stats of the default bench command, PMC passes on
# the shipped kernels (incl. the stream-K GEMM on the decode step's fused residual projections), and the one-setting-per-process
# re-measurement of the mid-batch "two forced K slices" knob.  Arg 1 = git revision for the records.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
GIT=${1:-unknown}
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r5_gpu_suite.log 2>&1
echo "GPU suite rc=$?"; tail -4 $O/r5_gpu_suite.log
# ---- rocprofv3 --kernel-trace --stats of the bench command (same flags as round 4's record) ----
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r5_prof_b4096" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r5_prof_b4096.log 2>&1
python tools/summarize_prof.py stats $O/r5_prof_b4096/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --no-ceilings (block_main_b4_1.2b P2048/D128, B = 4096), round 5 tree $GIT" --top 22 > $O/r5_bench_b4096_kernel_stats.md 2>&1
tail -1 $O/r5_prof_b4096.log | cut -c1-300 > $O/r5_bench_b4096_under_rocprof.json
head -40 $O/r5_bench_b4096_kernel_stats.md
rm -rf $O/r5_prof_b4096
# ---- PMC passes (separate --pmc runs, no trace domains) ----
SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
pass() {  # tag, counters, probe args...
  local tag=$1; local ctr=$2; shift 2
  ( cd /tmp && timeout 150 rocprofv3 --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/r5_pmc_$tag" -- python "$GRAFT_REPO_ROOT/tools/kernel_probe.py" --product "$@" ) > $O/r5_pmc_$tag.log 2>&1
}
for set in A F W; do
  case $set in A) C="$SQA";; F) C="FETCH_SIZE";; W) C="WRITE_SIZE";; esac
  pass gemm_plain_$set "$C" gemm --m 32768 --n 6144 --k 2048 --iters 5
  python tools/summarize_prof.py pmc $O/r5_pmc_gemm_plain_$set/*/ --kernel "gemm_bf16_mfma256pp2" > $O/r5_pmc_gemm_plain_$set.json 2>&1
done
python tools/summarize_prof.py traffic $O/r5_pmc_gemm_plain_F/*/ $O/r5_pmc_gemm_plain_W/*/ --kernel gemm_bf16_mfma256pp2 \
   --shape "M=32768 N=6144 K=2048 bf16 (prefill QKV shape, plain epilogue)" --algorithmic-bytes 562036736 --key linear_mfma \
   --kernel-name "gemm_bf16_mfma256pp2_kernel<0, 0> (persistent 256x256, second-generation K-loop)" --git $GIT --update $O/r5_pmc_traffic.json > /dev/null
pass gemm_dec_A "$SQA" gemm --m 4096 --n 6144 --k 2048 --iters 5
python tools/summarize_prof.py pmc $O/r5_pmc_gemm_dec_A/*/ --kernel "gemm_bf16_mfma256pp2" > $O/r5_pmc_gemm_dec_A.json 2>&1
pass sk_res_A "$SQA" sk --m 4096 --iters 5
python tools/summarize_prof.py pmc $O/r5_pmc_sk_res_A/*/ --kernel "gemm_bf16_mfma256sk" > $O/r5_pmc_sk_res_A.json 2>&1
python tools/summarize_prof.py pmc $O/r5_pmc_sk_res_A/*/ --kernel "gemm_bf16_mfma256x128" > $O/r5_pmc_t21_res_A.json 2>&1
for set in F W; do
  case $set in F) C="FETCH_SIZE";; W) C="WRITE_SIZE";; esac
  pass attn_$set "$C" attn --batch 1024 --keys 528 --iters 5
done
python tools/summarize_prof.py traffic $O/r5_pmc_attn_F/*/ $O/r5_pmc_attn_W/*/ --kernel attend_rows_kernel \
   --shape "B=1024, 16 heads, d=128, n=528 keys, bf16" --algorithmic-bytes 4429185024 --key attn_rows \
   --kernel-name "attend_rows_kernel<bf16, 128, 4, false, false>" --git $GIT --update $O/r5_pmc_traffic.json > /dev/null
rm -rf $O/r5_pmc_*/
cat $O/r5_pmc_traffic.json
for f in $O/r5_pmc_gemm_plain_A.json $O/r5_pmc_gemm_dec_A.json $O/r5_pmc_sk_res_A.json $O/r5_pmc_t21_res_A.json; do echo $f; grep -A10 derived $f | tr -d '\n '; echo; done
# ---- mid batch: bt_debug_mid_force2 on / off, ONE setting per process, processes alternated, medians of 12 generates ----
for rep in 1 2; do for B in 96 128 192; do for v in 1 0; do
  timeout 120 python tools/bench_knobs.py mid_force2=$v -- --batch $B --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'batch': $B, 'mid_force2': $v, 'rep': $rep, 'tok_s': round(d['value'],1), 'ms': round(d['ms_per_step'],3)}))"
done; done; done > $O/r5_ab_mid_force2_processes.jsonl
cat $O/r5_ab_mid_force2_processes.jsonl
