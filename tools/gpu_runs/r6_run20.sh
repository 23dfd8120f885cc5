#!/bin/bash
# round 6, GPU call 20: kernel stats at B = 256 (half mode of the fused mid-batch launch) and B = 512
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
GIT=$(cat .git_rev 2>/dev/null || echo a141d67)
for B in 256 512; do
  rm -rf $O/r6_prof_b$B
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r6_prof_b$B" -- python "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r6_prof_b$B.log 2>&1
  python tools/summarize_prof.py stats $O/r6_prof_b$B/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings (P2048/D128), round 6 tree $GIT" --top 16 > $O/r6_bench_b${B}_kernel_stats_fused.md 2>&1
  rm -rf $O/r6_prof_b$B
done
head -30 $O/r6_bench_b256_kernel_stats_fused.md | cut -c1-220
