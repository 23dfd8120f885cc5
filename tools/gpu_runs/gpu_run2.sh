#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_model_gpu.py tests/test_parallel_gpu.py -m gpu -q -p no:cacheprovider -x --deselect tests/test_model_gpu.py::test_embedder_g1 -k "real_dims or round or full or fp32_production or long_decode or pipeline or graphs_follow or topk or benchmark_printout or measure_generation or inference_demo or sampling or streamer or checkpoint or forward_scoring or two_rccl or small_batch or mid_batch or eos or shard" -s ) > gpurun_out/r2_pytest2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_pytest2.log
tail -8 gpurun_out/r2_pytest2.log
timeout 300 python tools/overlap_probe.py > gpurun_out/r2_overlap_probe.jsonl 2> gpurun_out/r2_overlap_probe.err
echo "probe rc=$?"; cat gpurun_out/r2_overlap_probe.jsonl
timeout 200 python tools/overlap_probe.py --rows 4096 > gpurun_out/r2_overlap_probe_m4096.jsonl 2>> gpurun_out/r2_overlap_probe.err
cat gpurun_out/r2_overlap_probe_m4096.jsonl
