#!/bin/bash
# round 4, GPU call 14: wide argmax (kernel tests + B = 1 / 8), PMC passes on the shipped kernels, rocprofv3 kernel stats of the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "argmax or sample" 2>&1 | tail -3
for b in 1 8; do
  timeout 300 python bench.py --batch $b --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-parity --no-secondary > $O/r4_bench_argmax_b$b.json 2> $O/r4_bench_argmax_b$b.log
  python -c "import json;d=json.load(open('$O/r4_bench_argmax_b$b.json'));print('B=$b', round(d['value'],1), 'tok/s')"
done
bash tools/gpu_runs/r4_pmc.sh $1 2>&1 | tail -25
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r4_prof_b4096" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary ) > $O/r4_prof_b4096.log 2>&1
python tools/summarize_prof.py stats $O/r4_prof_b4096/*/ --title "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary (default workload: B = 4096, P2048/D128), round 4" --top 16 > $O/r4_bench_b4096_kernel_stats.md 2>&1
rm -rf $O/r4_prof_b4096
head -30 $O/r4_bench_b4096_kernel_stats.md | cut -c1-180
grep "^{" $O/r4_prof_b4096.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_family_ms'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
