#!/bin/bash
# round 6, GPU call 25: timing-only ablations of the half-mode fused launch at B = 256 (what bounds its 1600-cycle K-tile?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for D in 0 1 2 4 8 16 18 22 3; do
  BT_MID_TL=1 timeout 300 python tools/bench_knobs.py mid_tall_dbg=$D -- --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings 2> $O/r6_mid_abl_$D.err | tail -1 > $O/r6_mid_abl_$D.json
  grep "bench_knobs" $O/r6_mid_abl_$D.err | sed 's/^\[bench_knobs\] //' > $O/r6_mid_abl_timeline_$D.json
  python - <<PY
import json
d = json.loads(open('$O/r6_mid_abl_timeline_$D.json').read())
b = json.loads(open('$O/r6_mid_abl_$D.json').read())
print('dbg', $D, 'g0', d['group0_median_ticks'], 'decode tok/s', round(b.get('decode_phase_tokens_per_s_per_gpu', 0)))
PY
done
