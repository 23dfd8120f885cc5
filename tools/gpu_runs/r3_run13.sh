#!/bin/bash
# round 3, GPU call 13: compaction of finished sequences (tokens == masked engine == oracle; speed-up), QKV epilogue row split,
# select-free GELU (accuracy test), EOS regression tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -s \
   -k "compaction or eos or gelu_epilogue or qkv_rope_append or ragged or streamer or g6_g9 or shard_equivalence or test_linear" ) > $O/r3_pytest13.log 2>&1
echo "pytest rc=$?" >> $O/r3_pytest13.log
grep -E "passed|failed|rc=|compaction\]|^E " $O/r3_pytest13.log | tail -12
timeout 300 python tools/kernel_probe.py ppab --variant 12 --ab-other 16 > $O/r3_ppab_gelu.jsonl 2> $O/r3_ppab.err
python -c "
import sys, json
for l in open('gpurun_out/r3_ppab_gelu.jsonl'):
    r = json.loads(l)
    if r['mode'] != 'res': print(r['M'], r['N'], r['K'], r['mode'], 'v2', min(r['a_us']), 'TF', r['a_TF'])"
