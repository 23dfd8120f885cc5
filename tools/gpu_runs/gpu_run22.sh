#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python bench.py > $O/r2_bench_default_final.json 2> $O/r2_bench_default_final.err
echo "bench default rc=$?"; python -c "import json;d=json.loads(open('$O/r2_bench_default_final.json').read().strip().splitlines()[-1]);print(round(d['value'],1),d['steps'],d['warmup'],round(d['prefill_ms'],1),round(d['decode_ms'],1),d['roofline']['frac'],d['roofline_decode_phase']['frac_of_binding_roof'],d['kernel_family_ms'],d['cpu_baseline']['value'])"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r2_prof_final" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-roofline --no-cpu-baseline ) > $O/r2_prof_final.log 2>&1
python tools/summarize_prof.py stats $O/r2_prof_final/*/ --title "bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline (B = 4096, P2048/D128; 2 generate() calls: 1 timed + 1 phase split), round 2 final tree" --top 16 > $O/r2_bench_b4096_kernel_stats_final.md 2>&1
head -24 $O/r2_bench_b4096_kernel_stats_final.md
rm -rf $O/r2_prof_final
timeout 400 python bench.py --steps 2 --warmup 1 --prefill 128 --decode 2048 --no-cpu-baseline > $O/r2_bench_p128_d2048_final.json 2> $O/r2_bench_p128_d2048_final.err
python -c "import json;d=json.loads(open('$O/r2_bench_p128_d2048_final.json').read().strip().splitlines()[-1]);print('p128/d2048',round(d['value']),d['config']['batch_per_gpu'],round(d['prefill_ms'],1),round(d['decode_ms'],1),d['roofline_decode_phase']['frac_of_binding_roof'])"
timeout 400 python bench.py --steps 2 --warmup 1 --config block_main_b8_1.2b --no-cpu-baseline --no-roofline > $O/r2_bench_b8_final.json 2> $O/r2_bench_b8_final.err
python -c "import json;d=json.loads(open('$O/r2_bench_b8_final.json').read().strip().splitlines()[-1]);print('L=8',round(d['value']),d['config']['batch_per_gpu'],round(d['prefill_ms'],1),round(d['decode_ms'],1))"
timeout 400 python bench.py --steps 1 --warmup 1 --prefill 2048 --decode 2048 --no-cpu-baseline --no-roofline > $O/r2_bench_p2048_d2048_final.json 2> $O/r2_bench_p2048_d2048_final.err
python -c "import json;d=json.loads(open('$O/r2_bench_p2048_d2048_final.json').read().strip().splitlines()[-1]);print('p2048/d2048',round(d['value']),d['config']['batch_per_gpu'],round(d['prefill_ms'],1),round(d['decode_ms'],1))"
for B in 1 2 4 8 16; do
  timeout 200 python bench.py --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r2_bench_small_final_b$B.json 2> $O/r2_bench_small_final_b$B.err
  python -c "import json;d=json.loads(open('$O/r2_bench_small_final_b$B.json').read().strip().splitlines()[-1]);print('B',$B,round(d['value'],1),round(d['prefill_ms'],1),round(d['decode_ms'],1))"
done
