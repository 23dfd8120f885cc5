#!/bin/bash
# round 5, GPU call 13: mid-batch defaults after calls 10-12 (256 x 128 slab kernel from 97 rows, write-through slabs there only), the
# interleaved QKV + attention + GELU reduce launch and the batched slab loads of the reduce kernels: parity tests, per-process A/B,
# kernel stats at B = 256.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
GIT=${1:-unknown}
O=gpurun_out
mkdir -p $O
timeout 700 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "tall or mid_rows or mid_batch" > $O/r5_run13_tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/r5_run13_tests.log
ab() {  # batch, knobs...
  local B=$1; shift
  timeout 150 python tools/bench_knobs.py "$@" -- --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'batch': $B, 'knobs': '$*', 'tok_s': round(d['value'],1), 'ms': round(d['ms_per_step'],3), 'prefill_ms': round(d['prefill_ms'],2), 'decode_ms': round(d['decode_ms'],2)}))"
}
{
for B in 256 192 128 64; do for v in 0 1; do ab $B mid_interleave=$v; done; done
ab 256 mid_tall=0
ab 256 mid_interleave=0
ab 256 mid_interleave=1
} > $O/r5_ab_mid_reduce_interleave.jsonl
cat $O/r5_ab_mid_reduce_interleave.jsonl
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r5_prof_b256" -- python "$GRAFT_REPO_ROOT/bench.py" --batch 256 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r5_prof_b256.log 2>&1
python tools/summarize_prof.py stats $O/r5_prof_b256/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --batch 256 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings (P2048/D128), round 5 tree $GIT" --top 18 > $O/r5_bench_b256_kernel_stats.md 2>&1
rm -rf $O/r5_prof_b256
cut -c1-170 $O/r5_bench_b256_kernel_stats.md | sed -n 5,18p
