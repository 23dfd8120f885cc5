#!/bin/bash
# round 5, GPU call 14: the slab path up to 512 rows (two row tiles of the 256 x 128 slab kernel; bt_debug_mid_rows_max 256 = round-4 limit):
# parity tests, per-process A/B of the end-to-end bench at B = 160 ... 512.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 800 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "tall or mid_rows or mid_batch or rejects_cache_overflow" > $O/r5_run14_tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/r5_run14_tests.log
ab() {  # batch, knobs...
  local B=$1; shift
  timeout 150 python tools/bench_knobs.py "$@" -- --batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'batch': $B, 'knobs': '$*', 'tok_s': round(d['value'],1), 'ms': round(d['ms_per_step'],3), 'prefill_ms': round(d['prefill_ms'],2), 'decode_ms': round(d['decode_ms'],2)}))"
}
{
for B in 256 192 160 384 512; do for v in 256 512; do ab $B mid_rows_max=$v; done; done
ab 256 mid_rows_max=256
ab 256 mid_rows_max=512
} > $O/r5_ab_mid_rows_max.jsonl
cat $O/r5_ab_mid_rows_max.jsonl
