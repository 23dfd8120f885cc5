#!/bin/bash
# round 3, GPU call 3: second-generation K-loop of the persistent GEMM: per-op tests vs fp64 (all persistent-kernel tests), A/B vs
# the first generation on every bench shape and epilogue mode, bit-identity, timing ablations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x \
   -k "persistent or qkv_rope_append or test_linear or gelu_epilogue or 256_tile" ) > $O/r3_pytest3.log 2>&1
echo "pytest rc=$?" >> $O/r3_pytest3.log
tail -5 $O/r3_pytest3.log
timeout 300 python tools/kernel_probe.py ppv2 > $O/r3_ppv2.jsonl 2> $O/r3_ppv2.err
cat $O/r3_ppv2.jsonl
tail -3 $O/r3_ppv2.err
