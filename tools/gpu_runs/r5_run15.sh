#!/bin/bash
# round 5, GPU call 15: final-tree measurements of the mid-batch work with the PRODUCT library: batch sweep (B = 256 and 512 with the in-run
# oracle check of the timed batch), rocprofv3 kernel stats at B = 256, and a short default-batch run (B = 4096) as a regression check.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
GIT=${1:-unknown}
O=gpurun_out
mkdir -p $O
for B in 256 512; do
  timeout 200 python bench.py --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-ceilings 2>/dev/null | tail -1 > $O/r5_sweep_final_b$B.json
done
for B in 8 64 96 128 160 192 384 1024; do
  timeout 150 python bench.py --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | tail -1 > $O/r5_sweep_final_b$B.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5_sweep_final_b*.json'), key=lambda s: int(s.split('_b')[-1].split('.')[0])):
    try:
        d = json.loads(open(f).read())
        print(f.split('/')[-1], round(d['value'], 1), 'decode-phase', round(d.get('decode_phase_tokens_per_s_per_gpu', 0), 1), 'prefill', round(d['prefill_ms'], 1), 'decode', round(d['decode_ms'], 1), 'parity', (d.get('parity_in_run') or {}).get('ok'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r5_prof_b256" -- python "$GRAFT_REPO_ROOT/bench.py" --batch 256 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r5_prof_b256.log 2>&1
python tools/summarize_prof.py stats $O/r5_prof_b256/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --batch 256 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings (P2048/D128), round 5 final tree $GIT" --top 18 > $O/r5_bench_b256_kernel_stats.md 2>&1
rm -rf $O/r5_prof_b256
cut -c1-170 $O/r5_bench_b256_kernel_stats.md | sed -n 3,18p
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/r5_bench_default_after_mid_batch.json
python -c "
import json; d=json.loads(open('$O/r5_bench_default_after_mid_batch.json').read()); print('default', round(d['value'],1), d['ms_per_step'], (d.get('parity_in_run') or {}).get('ok'), d['roofline']['frac'])"
