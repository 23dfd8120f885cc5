#!/bin/bash
# round 3, GPU call 18 (final tree): full -m gpu suite, smoke, default bench line, rocprofv3 kernel stats of the bench command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/r3_pytest_final.log 2>&1
echo "pytest rc=$?" >> $O/r3_pytest_final.log
tail -8 $O/r3_pytest_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/r3_smoke_final.log 2>&1; tail -2 $O/r3_smoke_final.log
timeout 600 python bench.py > $O/r3_bench_default_final.json 2> $O/r3_bench_default_final.err
cut -c1-1500 $O/r3_bench_default_final.json; tail -5 $O/r3_bench_default_final.err
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r3_prof_final" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-roofline --no-cpu-baseline ) > $O/r3_prof_final.log 2>&1
python tools/summarize_prof.py stats $O/r3_prof_final/*/ --title "bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline (B = 4096, P2048/D128; 2 generate() calls: 1 timed + 1 phase split), round 3 final tree" --top 20 > $O/r3_bench_b4096_kernel_stats_final.md 2>&1
head -32 $O/r3_bench_b4096_kernel_stats_final.md | cut -c1-200
rm -rf $O/r3_prof_final
