#!/bin/bash
# round 6, GPU call 11: epilogue with the two waves of a SIMD alternating at raised priority (variant 97) vs production
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 500 python tools/kernel_probe.py ppab --variant 12 --ab-other 97 > $O/r6_ppab_epilogue_prio.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r6_ppab_epilogue_prio.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    if d.get('op') == 'pp_ab': print('production (a) vs alternating priority (b)', d['mode'], d['M'], d['N'], d['K'], d['a_us'], d['b_us'], d['a_TF'], d['b_TF'], d['bit_identical'])
PY
timeout 300 python tools/kernel_probe.py ppqkv --variant 12 --ab-other 97 > $O/r6_ppqkv_epilogue_prio.jsonl 2>/dev/null
cut -c1-300 $O/r6_ppqkv_epilogue_prio.jsonl
