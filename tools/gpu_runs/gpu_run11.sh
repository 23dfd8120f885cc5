#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 > $O/r2_bench_default.json 2> $O/r2_bench_default.err
echo "bench default rc=$?"; tail -c 1500 $O/r2_bench_default.json; tail -5 $O/r2_bench_default.err
timeout 400 python bench.py --steps 2 --warmup 1 --prefill 128 --decode 2048 --no-cpu-baseline > $O/r2_bench_p128_d2048.json 2> $O/r2_bench_p128_d2048.err
echo "bench p128/d2048 rc=$?"; python -c "import json;d=json.loads(open('$O/r2_bench_p128_d2048.json').read().strip().splitlines()[-1]);print(round(d['value']),d['config']['batch_per_gpu'],round(d['prefill_ms'],1),round(d['decode_ms'],1),d['roofline_decode_phase']['frac_of_binding_roof'],d['roofline_decode_phase']['binding'])"
timeout 400 python bench.py --steps 2 --warmup 1 --config block_main_b8_1.2b --no-cpu-baseline > $O/r2_bench_b8.json 2> $O/r2_bench_b8.err
echo "bench L=8 rc=$?"; python -c "import json;d=json.loads(open('$O/r2_bench_b8.json').read().strip().splitlines()[-1]);print(round(d['value']),d['config']['batch_per_gpu'],round(d['prefill_ms'],1),round(d['decode_ms'],1))"
timeout 400 python bench.py --steps 1 --warmup 1 --prefill 2048 --decode 2048 --no-cpu-baseline --no-roofline > $O/r2_bench_p2048_d2048.json 2> $O/r2_bench_p2048_d2048.err
echo "bench p2048/d2048 (configs[3] per-GPU) rc=$?"; python -c "import json;d=json.loads(open('$O/r2_bench_p2048_d2048.json').read().strip().splitlines()[-1]);print(round(d['value']),d['config']['batch_per_gpu'],round(d['prefill_ms'],1),round(d['decode_ms'],1))"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r2_prof_b4096" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-roofline --no-cpu-baseline ) > $O/r2_prof_b4096.log 2>&1
python tools/summarize_prof.py stats $O/r2_prof_b4096/*/ --title "bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline (B = 4096, P2048/D128; 2 generate() calls: 1 timed + 1 phase split), round 2" --top 16 > $O/r2_bench_b4096_kernel_stats.md 2>&1
head -30 $O/r2_bench_b4096_kernel_stats.md
