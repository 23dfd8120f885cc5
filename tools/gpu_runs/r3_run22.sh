#!/bin/bash
# round 3, GPU call 22: experiment -- the last 4 MFMAs of a cluster issued after the hand-over barrier (variant 19) vs production
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
V=${1:-19}
timeout 400 python tools/kernel_probe.py ppab --variant 12 --ab-other $V > $O/r3_ppab_v$V.jsonl 2> $O/r3_ppab_v$V.err
python -c "
import json
for l in open('gpurun_out/r3_ppab_v$V.jsonl'):
    r = json.loads(l); print(r['M'], r['N'], r['K'], r['mode'], 'prod', min(r['a_us']), 'other', min(r['b_us']), 'x%.3f' % (min(r['a_us'])/min(r['b_us'])), 'TF', r['b_TF'], 'same' if r['bit_identical'] else 'DIFFERENT')"
tail -3 $O/r3_ppab_v$V.err
