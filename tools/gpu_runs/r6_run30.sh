#!/bin/bash
# round 6, GPU call 30: final tree -- smoke() as the driver runs it, the whole GPU suite, the default no-flag bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $O/r6_smoke_final.log 2>&1
echo "smoke rc=$?"; tail -2 $O/r6_smoke_final.log | cut -c1-200
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > $O/r6_gpu_suite.log 2>&1
echo "GPU suite rc=$?"; tail -22 $O/r6_gpu_suite.log | cut -c1-200
( time timeout 900 python bench.py > $O/r6_bench_noflags.json 2> $O/r6_bench_noflags.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6_bench_noflags.json').read().strip().splitlines()[-1])
print('no-flag bench', round(d['value'], 1), d['steps'], d['warmup'], d['roofline']['frac'], (d.get('parity_in_run') or {}).get('ok'), [round(s['value']) for s in d.get('secondary', [])])
PY
