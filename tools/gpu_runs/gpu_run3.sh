#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 600 python -m pytest tests/test_model_gpu.py tests/test_parallel_gpu.py -m gpu -q -p no:cacheprovider -k "pipeline or graphs_follow or topk or benchmark_printout or measure_generation or inference_demo or two_rccl or real_dims" ) > $O/r2_pytest3.log 2>&1
echo "pytest rc=$?" >> $O/r2_pytest3.log; tail -6 $O/r2_pytest3.log
for R in 8192 16384 32768 65536; do
  BT_PREFILL_ROWS=$R timeout 200 python bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline > $O/r2_prefill_rows_$R.json 2> $O/r2_prefill_rows_$R.err
  python -c "import json;d=json.loads(open('$O/r2_prefill_rows_$R.json').read().strip().splitlines()[-1]);print('prefill_rows',$R,round(d['prefill_ms'],1),round(d['decode_ms'],1),round(d['value']))"
done
for B in 64 256; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r2_prof_b$B" -- python "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 1 --warmup 1 --no-roofline --no-cpu-baseline ) > $O/r2_prof_b$B.log 2>&1
  python tools/summarize_prof.py stats $O/r2_prof_b$B/*/ --title "bench.py --batch $B (P2048/D128), 3 generate() calls" --top 24 > $O/r2_bench_b${B}_kernel_stats.md 2>&1
  head -32 $O/r2_bench_b${B}_kernel_stats.md
  grep -o '"value": [0-9.]*\|"decode_ms": [0-9.]*' $O/r2_prof_b$B.log | tr '\n' ' '; echo
done
# local-attention PMC evidence (separate --pmc passes, no trace domains)
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | cut -d' ' -f1)
  ( cd /tmp && timeout 120 rocprofv3 --pmc $C -d "$GRAFT_REPO_ROOT/$O/r2_pmc_local_$tag" -- python "$GRAFT_REPO_ROOT/tools/kernel_probe.py" local --batch 4096 --iters 5 ) > $O/r2_pmc_local_$tag.log 2>&1
  python tools/summarize_prof.py pmc $O/r2_pmc_local_$tag/*/ --kernel "attend_rows_kernel" > $O/r2_pmc_local_$tag.json 2>&1; cat $O/r2_pmc_local_$tag.json | head -30
done
python tools/kernel_probe.py local --batch 4096 --iters 20 > $O/r2_probe_local.jsonl 2>/dev/null; cat $O/r2_probe_local.jsonl
