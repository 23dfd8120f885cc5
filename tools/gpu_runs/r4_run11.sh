#!/bin/bash
# round 4, GPU call 11: rocprofv3 kernel stats of the small / mid batch paths on the round-4 dispatch (B = 1, 8, 64, 256)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for B in 1 8 64 256; do
  ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r4_prof_b$B" -- python "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary ) > $O/r4_prof_b$B.log 2>&1
  python tools/summarize_prof.py stats $O/r4_prof_b$B/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --batch $B --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary (P2048/D128), round 4" --top 18 > $O/r4_bench_b${B}_kernel_stats.md 2>&1
  rm -rf $O/r4_prof_b$B
  head -26 $O/r4_bench_b${B}_kernel_stats.md
done
