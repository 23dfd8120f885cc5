#!/bin/bash
# round 5, GPU call 12: write-through (sc1) slab stores -- the 256 x 128 slab kernel (bt_debug_mid_tall_wt) and the 64 x 64 slab kernels
# (bt_debug_mid_wt) -- parity tests, per-process A/B of the end-to-end bench, workgroup budget of the residual launch, kernel stats at B = 256.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
GIT=${1:-unknown}
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "tall or mid_rows_decoder_stack or mid_batch_generate or mid_batch_dual or in_launch_reduction" > $O/r5_run12_tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/r5_run12_tests.log
ab() {  # batch, knobs...
  local B=$1; shift
  timeout 150 python tools/bench_knobs.py "$@" -- --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'batch': $B, 'knobs': '$*', 'tok_s': round(d['value'],1), 'ms': round(d['ms_per_step'],3), 'prefill_ms': round(d['prefill_ms'],2), 'decode_ms': round(d['decode_ms'],2)}))"
}
{
ab 256 mid_tall=1 mid_tall_wt=0
ab 256 mid_tall=1 mid_tall_wt=1
ab 256 mid_tall=1 mid_tall_wt=1 mid_tall_wgs_res=128
ab 256 mid_tall=0 mid_wt=0
ab 256 mid_tall=0 mid_wt=1
ab 192 mid_tall=1 mid_tall_wt=0
ab 192 mid_tall=1 mid_tall_wt=1
for B in 128 96 64 32; do for v in 0 1; do ab $B mid_wt=$v; done; done
ab 128 mid_tall=2
ab 96 mid_tall=2
ab 256 mid_tall=1 mid_tall_wt=0
ab 256 mid_tall=1 mid_tall_wt=1
} > $O/r5_ab_mid_wt.jsonl
cat $O/r5_ab_mid_wt.jsonl
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r5_prof_b256" -- python "$GRAFT_REPO_ROOT/bench.py" --batch 256 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r5_prof_b256.log 2>&1
python tools/summarize_prof.py stats $O/r5_prof_b256/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --batch 256 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings (P2048/D128), round 5 tree $GIT" --top 18 > $O/r5_bench_b256_kernel_stats.md 2>&1
rm -rf $O/r5_prof_b256
cut -c1-170 $O/r5_bench_b256_kernel_stats.md | sed -n 5,16p
