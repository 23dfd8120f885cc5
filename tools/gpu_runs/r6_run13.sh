#!/bin/bash
# round 6, GPU call 13: does the half-batch pipelined engine pay once the persistent GEMM leaves CUs to the other shard's attention?
# (bt_debug_gemm_max_wg caps the persistent kernel's grid; the stream-K residual GEMM keeps all CUs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
FLAGS="--steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-parity --no-roofline --no-ceilings"
timeout 300 python tools/bench_knobs.py gemm_max_wg=0 -- $FLAGS > $O/r6_pipe_single_wg0.json 2>/dev/null
for wg in 0 224 192 160; do
  BT_ENGINE=pipelined timeout 300 python tools/bench_knobs.py gemm_max_wg=$wg -- $FLAGS > $O/r6_pipe_two_shards_wg$wg.json 2> $O/r6_pipe_two_shards_wg$wg.err
done
timeout 300 python tools/bench_knobs.py gemm_max_wg=192 -- $FLAGS > $O/r6_pipe_single_wg192.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6_pipe_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), 'tok/s prefill', round(d['prefill_ms']), 'decode', round(d['decode_ms']))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 $O/r6_pipe_two_shards_wg192.err
