#!/bin/bash
# round 3, GPU call 27: descriptor-addressed K / V stores of the fused QKV epilogue: tests (debug + product library) and A/B vs the
# generic addresses (variant 27)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "qkv" ) > $O/r3_pytest27.log 2>&1
tail -4 $O/r3_pytest27.log
rm -f $O/r3_ppqkv_ab.jsonl
for V in 27 27; do
  timeout 300 python tools/kernel_probe.py ppqkv --variant 12 --ab-other $V >> $O/r3_ppqkv_ab.jsonl 2> $O/r3_ppqkv_ab.err
done
cut -c1-330 $O/r3_ppqkv_ab.jsonl; tail -2 $O/r3_ppqkv_ab.err
( time timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x -k "g6_g9 or full_depth or long_prompt or eos or ragged" ) > $O/r3_pytest27b.log 2>&1
tail -4 $O/r3_pytest27b.log
