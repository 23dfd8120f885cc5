#!/bin/bash
# round 6, GPU call 36: kernel stats of the decode-heavy secondary workload (configs[2], P128/D2048 at B = 4096) on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -rf $O/r6_prof_p128
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r6_prof_p128" -- python "$GRAFT_REPO_ROOT/bench.py" --prefill 128 --decode 2048 --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r6_prof_p128.log 2>&1
echo "rc=$?"; tail -1 $O/r6_prof_p128.log | cut -c1-300
python tools/summarize_prof.py stats $O/r6_prof_p128/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --prefill 128 --decode 2048 --steps 1 --warmup 1 (block_main_b4_1.2b, B = 4096, configs[2]), final round-6 tree" --top 16 > $O/r6_bench_p128_d2048_kernel_stats.md 2>&1
rm -rf $O/r6_prof_p128
head -26 $O/r6_bench_p128_d2048_kernel_stats.md | cut -c1-200
