#!/bin/bash
# round 3, GPU call 6: v2 kernel with descriptor-addressed stores / residual loads and the boundary half-tiles issued ahead of
# the stores: per-op tests vs fp64, A/B vs first-generation K-loop (which shares the new epilogue), epilogue parts
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x \
   -k "persistent or qkv_rope_append or test_linear or gelu_epilogue or 256_tile" ) > $O/r3_pytest6.log 2>&1
echo "pytest rc=$?" >> $O/r3_pytest6.log
tail -4 $O/r3_pytest6.log
timeout 300 python tools/kernel_probe.py ppv2 > $O/r3_ppv2b.jsonl 2> $O/r3_ppv2b.err
grep pp_v1_vs_v2 $O/r3_ppv2b.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['M'], r['N'], r['K'], r['mode'], 'v1', min(r['v1_us']), 'v2', min(r['v2_us']), 'TF', r['v2_TF'], 'same bits', r['bit_identical'])"
tail -2 $O/r3_ppv2b.err
