#!/bin/bash
# round 6, GPU call 23: timeline of the fused mid-batch launch (s_memtime stamps, debug build) at B = 256 (half mode) and B = 512 (two row tiles)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for B in 256 512 192; do
  BT_MID_TL=1 timeout 300 python tools/bench_knobs.py -- --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2> $O/r6_mid_tl_b$B.err | tail -1 > $O/r6_mid_tl_b$B.json
  grep "bench_knobs" $O/r6_mid_tl_b$B.err | sed 's/^\[bench_knobs\] //' > $O/r6_mid_fused_timeline_b$B.json
  cat $O/r6_mid_fused_timeline_b$B.json
done
