#!/bin/bash
# round 3, GPU call 7: one-batch residual loads; A/B of the DMA-inside-the-MFMA-cluster variant (17) vs production (12)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x \
   -k "persistent or qkv_rope_append_fused_persistent" ) > $O/r3_pytest7.log 2>&1
echo "pytest rc=$?" >> $O/r3_pytest7.log
tail -4 $O/r3_pytest7.log
timeout 300 python tools/kernel_probe.py ppab --variant 12 --ab-other 17 > $O/r3_ppab_dmac.jsonl 2> $O/r3_ppab.err
python -c "
import sys, json
for l in open('gpurun_out/r3_ppab_dmac.jsonl'):
    r = json.loads(l)
    print(r['M'], r['N'], r['K'], r['mode'], 'prod', min(r['a_us']), 'dmac', min(r['b_us']), 'TF', r['a_TF'], r['b_TF'], 'same bits', r['bit_identical'])"
tail -2 $O/r3_ppab.err
