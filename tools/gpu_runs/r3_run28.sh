#!/bin/bash
# round 3, GPU call 28 (final tree): rocprofv3 kernel stats of the small / mid batch paths (unchanged this round; evidence refresh)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for B in 1 64; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r3_prof_b$B" -- python "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline ) > $O/r3_prof_b$B.log 2>&1
  python tools/summarize_prof.py stats $O/r3_prof_b$B/*/ --title "bench.py --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline (P2048/D128), round 3 final tree" --top 16 > $O/r3_bench_b${B}_kernel_stats_final.md 2>&1
  head -24 $O/r3_bench_b${B}_kernel_stats_final.md | cut -c1-170
  rm -rf $O/r3_prof_b$B
done
