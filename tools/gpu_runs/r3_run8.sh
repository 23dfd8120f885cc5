#!/bin/bash
# round 3, GPU call 8: residual joined in the store layout (full-line loads); tests + per-mode timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x \
   -k "persistent or test_linear or 256_tile or strided" ) > $O/r3_pytest8.log 2>&1
echo "pytest rc=$?" >> $O/r3_pytest8.log
tail -4 $O/r3_pytest8.log
timeout 300 python tools/kernel_probe.py ppab --variant 12 --ab-other 16 > $O/r3_ppab_res.jsonl 2> $O/r3_ppab.err
python -c "
import sys, json
for l in open('gpurun_out/r3_ppab_res.jsonl'):
    r = json.loads(l)
    print(r['M'], r['N'], r['K'], r['mode'], 'v2', min(r['a_us']), 'v1', min(r['b_us']), 'TF', r['a_TF'], r['b_TF'], 'same bits', r['bit_identical'])"
tail -2 $O/r3_ppab.err
