#!/bin/bash
# round 6, GPU call 29: evidence on the FINAL round-6 tree (the GPU suite ran in call 28) -- the default bench line as the driver runs it (headline +
# three secondary legs + parity + CPU baseline), rocprofv3 kernel stats of the bench command, PMC passes on the shipped GEMMs
# (separate --pmc runs, no trace domains), batch sweep.  Arg 1 = git revision for the records.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
GIT=${1:-unknown}
O=gpurun_out
mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r6_bench_default_final.json 2> $O/r6_bench_default_final.err
echo "bench rc=$?"; grep "secondary\|parity\|timed steps\|cpu" $O/r6_bench_default_final.err | tail -14
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6_bench_default_final.json').read().strip().splitlines()[-1])
print('default', round(d['value'], 1), d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_sustained'), d['kernel_family_ms'], (d.get('parity_in_run') or {}).get('ok'))
for s in d.get('secondary', []): print('  secondary', s['workload'][:60], s['batch_per_gpu'], round(s['value']), s['ms_each_step'], (s.get('parity_in_run') or {}).get('ok'), s['roofline_decode_phase']['frac_of_binding_roof'])
print('  decode phase', d.get('roofline_decode_phase'))
print('  ceilings', {k: d['ceilings'][k] for k in ('copy_GBs', 'mfma_sustained_TFLOPs', 'sclk_MHz_mean', 'power_W_mean', 'power_W_max')})
PY
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r6_prof_b4096" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r6_prof_b4096.log 2>&1
python tools/summarize_prof.py stats $O/r6_prof_b4096/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --no-ceilings (block_main_b4_1.2b P2048/D128, B = 4096), round 6 tree $GIT" --top 22 > $O/r6_bench_b4096_kernel_stats.md 2>&1
tail -1 $O/r6_prof_b4096.log | cut -c1-300 > $O/r6_bench_b4096_under_rocprof.json
head -16 $O/r6_bench_b4096_kernel_stats.md | cut -c1-200
rm -rf $O/r6_prof_b4096
SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
pass() {  # tag, counters, probe args...
  local tag=$1; local ctr=$2; shift 2
  ( cd /tmp && timeout 150 rocprofv3 --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/r6_pmc_$tag" -- python "$GRAFT_REPO_ROOT/tools/kernel_probe.py" --product "$@" ) > $O/r6_pmc_$tag.log 2>&1
}
for set in A F W; do
  case $set in A) C="$SQA";; F) C="FETCH_SIZE";; W) C="WRITE_SIZE";; esac
  pass gemm_plain_$set "$C" gemm --m 32768 --n 6144 --k 2048 --iters 5
  python tools/summarize_prof.py pmc $O/r6_pmc_gemm_plain_$set/*/ --kernel "gemm_bf16_mfma256pp2" > $O/r6_pmc_gemm_plain_$set.json 2>&1
done
python tools/summarize_prof.py traffic $O/r6_pmc_gemm_plain_F/*/ $O/r6_pmc_gemm_plain_W/*/ --kernel gemm_bf16_mfma256pp2 \
   --shape "M=32768 N=6144 K=2048 bf16 (prefill QKV shape, plain epilogue)" --algorithmic-bytes 562036736 --key linear_mfma \
   --kernel-name "gemm_bf16_mfma256pp2_kernel<0, 0> (persistent 256x256, round-6 tile boundary)" --git $GIT --update $O/r6_pmc_traffic.json > /dev/null
pass gemm_gelu_A "$SQA" gemm --m 32768 --n 8192 --k 2048 --act gelu --iters 5
python tools/summarize_prof.py pmc $O/r6_pmc_gemm_gelu_A/*/ --kernel "gemm_bf16_mfma256pp2" > $O/r6_pmc_gemm_gelu_A.json 2>&1
pass gemm_dec_A "$SQA" gemm --m 4096 --n 6144 --k 2048 --iters 5
python tools/summarize_prof.py pmc $O/r6_pmc_gemm_dec_A/*/ --kernel "gemm_bf16_mfma256pp2" > $O/r6_pmc_gemm_dec_A.json 2>&1
pass sk_res_A "$SQA" sk --m 4096 --iters 5
python tools/summarize_prof.py pmc $O/r6_pmc_sk_res_A/*/ --kernel "gemm_bf16_mfma256sk" > $O/r6_pmc_sk_res_A.json 2>&1
for set in F W; do
  case $set in F) C="FETCH_SIZE";; W) C="WRITE_SIZE";; esac
  pass attn_$set "$C" attn --batch 1024 --keys 528 --iters 5
done
python tools/summarize_prof.py traffic $O/r6_pmc_attn_F/*/ $O/r6_pmc_attn_W/*/ --kernel attend_rows_kernel \
   --shape "B=1024, 16 heads, d=128, n=528 keys, bf16" --algorithmic-bytes 4429185024 --key attn_rows \
   --kernel-name "attend_rows_kernel<bf16, 128, 4, false, false>" --git $GIT --update $O/r6_pmc_traffic.json > /dev/null
rm -rf $O/r6_pmc_*/
cat $O/r6_pmc_traffic.json | tr -d '\n ' | cut -c1-600; echo
for f in plain gelu dec; do echo $f; grep -A8 derived $O/r6_pmc_gemm_${f}_A.json | tr -d '\n '; echo; done
echo sk; grep -A8 derived $O/r6_pmc_sk_res_A.json | tr -d '\n '; echo
for B in 1 8 64 128 192 256 384 512 1024; do
  timeout 200 python bench.py --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | tail -1 > $O/r6_sweep_b$B.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6_sweep_b*.json'), key=lambda s: int(s.split('_b')[-1].split('.')[0])):
    try:
        d = json.loads(open(f).read())
        print(f.split('/')[-1], round(d['value'], 1), 'decode-phase', round(d.get('decode_phase_tokens_per_s_per_gpu', 0), 1))
    except Exception as e:
        print(f, 'unreadable', e)
PY
