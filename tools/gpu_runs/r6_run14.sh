#!/bin/bash
# round 6, GPU call 14: memset-node probe; stream-K tests + a short headline run after the sk_plan memo (host-only change)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 200 python tools/memset_graph_probe.py > $O/r6_memset_graph_probe.jsonl 2>/dev/null; cat $O/r6_memset_graph_probe.jsonl
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "stream_k or handoff or large_stack or benchmark_batch" 2>&1 | tail -3
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/r6_bench_after_memo.json
python -c "
import json; d=json.loads(open('$O/r6_bench_after_memo.json').read()); print('default', round(d['value'],1), d['ms_per_step'], (d.get('parity_in_run') or {}).get('ok'), d['roofline']['frac'])"
