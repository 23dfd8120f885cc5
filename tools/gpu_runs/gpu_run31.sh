#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
for B in 1 64; do
  ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r2_prof3_b$B" -- python "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 1 --warmup 1 --no-roofline --no-cpu-baseline ) > $O/r2_prof3_b$B.log 2>&1
  python tools/summarize_prof.py stats $O/r2_prof3_b$B/*/ --title "bench.py --batch $B (P2048/D128), 3 generate() calls, round 2 final tree" --top 14 > $O/r2_bench_b${B}_kernel_stats_final.md 2>&1
  head -22 $O/r2_bench_b${B}_kernel_stats_final.md
  rm -rf $O/r2_prof3_b$B
done
