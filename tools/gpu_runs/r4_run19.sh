#!/bin/bash
# round 4, GPU call 19: persistent [QKV | h->4h] GEMV for <= 2 rows: bit-identity, same-box A/B at B = 1 / 2, kernel durations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "persistent_gemv or few_rows_dual or small_batch_fused or unwritten_cache" 2>&1 | tail -4
timeout 500 python tools/ab_small_batch.py --knob few_persist --batches 1,2 --reps 4 > $O/r4_ab_few_persist.jsonl 2> $O/r4_ab_few_persist.log
cat $O/r4_ab_few_persist.jsonl; tail -2 $O/r4_ab_few_persist.log
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r4_prof_b1p" -- python "$GRAFT_REPO_ROOT/bench.py" --batch 1 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary ) > $O/r4_prof_b1p.log 2>&1
python tools/summarize_prof.py stats $O/r4_prof_b1p/*/ --title "bench.py --batch 1 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary (P2048/D128), round 4, persistent [QKV | h->4h] GEMV" --top 10 > $O/r4_bench_b1_kernel_stats_persist.md 2>&1
rm -rf $O/r4_prof_b1p
head -16 $O/r4_bench_b1_kernel_stats_persist.md | cut -c1-170
