#!/bin/bash
# round 5, GPU call 2: the stream-K items GEMM (gemm_bf16_mfma256sk_kernel): kernel tests vs fp64 / bit-identity / under load, the
# interleaved same-box A/B on the decode step's shapes, the model-level parity tests that run it through the engine, and a
# one-setting-per-process end-to-end A/B (debug library, knobs off vs on) on configs[1] and configs[2]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "stream_k" > $O/r5_run2_sk_tests.log 2>&1
RC=$?
echo "stream-K kernel tests rc=$RC"; tail -15 $O/r5_run2_sk_tests.log
timeout 300 python tools/kernel_probe.py sk --iters 30 > $O/r5_decode_shapes_sk.jsonl 2> $O/r5_decode_shapes_sk.log
cat $O/r5_decode_shapes_sk.jsonl; tail -3 $O/r5_decode_shapes_sk.log
if [ $RC -ne 0 ]; then echo "kernel tests failed: skipping model tests and bench"; exit 0; fi
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "linear2 or qkv_rope_append_product or persistent_inplace" > $O/r5_run2_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -4 $O/r5_run2_kernel_tests.log
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -s -m gpu -k "full_depth or two_branch or half_batch or streamer or large_stack or real_dims or full_size" > $O/r5_run2_model_tests.log 2>&1
RC2=$?
echo "model tests rc=$RC2"; grep -E "^\[|passed|failed|Error" $O/r5_run2_model_tests.log | tail -30
for cfg in "0 0" "1 1" "2 1" "1 2"; do
  set -- $cfg
  timeout 300 python tools/bench_knobs.py stream_k=$1 lin2_sk=$2 -- --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings > $O/r5_ab_sk$1$2_p2048.json 2> $O/r5_ab_sk$1$2_p2048.log
done
for cfg in "0 0" "1 1"; do
  set -- $cfg
  timeout 400 python tools/bench_knobs.py stream_k=$1 lin2_sk=$2 -- --prefill 128 --decode 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings > $O/r5_ab_sk$1$2_p128.json 2> $O/r5_ab_sk$1$2_p128.log
done
python - <<'PY'
import json
for name in ("r5_ab_sk00_p2048", "r5_ab_sk11_p2048", "r5_ab_sk21_p2048", "r5_ab_sk12_p2048", "r5_ab_sk00_p128", "r5_ab_sk11_p128"):
    try:
        d = json.load(open(f"gpurun_out/{name}.json"))
        print(name, round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms", "decode_ms", round(d.get("decode_ms", 0), 1), "prefill_ms", round(d.get("prefill_ms", 0), 1))
    except Exception as e:
        print(name, "ERR", e)
PY
