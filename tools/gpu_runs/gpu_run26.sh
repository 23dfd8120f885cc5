#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_parallel_gpu.py -m gpu -q -p no:cacheprovider -x ) 2>&1 | tail -3
( timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "attention" ) 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/r2_bench_default_final2.json 2> $O/r2_bench_default_final2.err
echo "bench default rc=$?"; python -c "import json;d=json.loads(open('$O/r2_bench_default_final2.json').read().strip().splitlines()[-1]);print(round(d['value'],1),d['steps'],d['warmup'],round(d['prefill_ms'],1),round(d['decode_ms'],1),d['roofline']['frac'],d['roofline_decode_phase']['frac_of_binding_roof'],d['kernel_family_ms'],d['cpu_baseline']['value'])"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r2_prof_final2" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-roofline --no-cpu-baseline ) > $O/r2_prof_final2.log 2>&1
python tools/summarize_prof.py stats $O/r2_prof_final2/*/ --title "bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline (B = 4096, P2048/D128; 2 generate() calls: 1 timed + 1 phase split), round 2 final tree" --top 16 > $O/r2_bench_b4096_kernel_stats_final.md 2>&1
head -16 $O/r2_bench_b4096_kernel_stats_final.md
rm -rf $O/r2_prof_final2
