#!/bin/bash
# round 5, GPU call 7: symmetric hand-off of the stream-K GEMM (tiles that are exactly two one-chain workgroups): kernel tests, same-box
# A/B (symmetric on / off, stream-K vs round-4 dispatch), model parity tests, the default bench line on the product library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "stream_k or linear2" > $O/r5_run5_sk_tests.log 2>&1
RC=$?
echo "stream-K + linear2 kernel tests rc=$RC"; tail -6 $O/r5_run5_sk_tests.log
timeout 400 python tools/kernel_probe.py sk --iters 30 > $O/r5_decode_shapes_sk_sym.jsonl 2> $O/r5_decode_shapes_sk_sym.log
cat $O/r5_decode_shapes_sk_sym.jsonl; tail -2 $O/r5_decode_shapes_sk_sym.log
if [ $RC -ne 0 ]; then echo "kernel tests failed: skipping the rest"; exit 0; fi
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -s -m gpu -k "full_depth_1p2b_bf16 or large_stack or full_size_1p2b_generate" > $O/r5_run5_model_tests.log 2>&1
echo "model tests rc=$?"; grep -E "^\[|passed|failed|Error" $O/r5_run5_model_tests.log | tail -14
timeout 900 python bench.py --steps 5 --warmup 2 > $O/r5_bench_default_2.json 2> $O/r5_bench_default_2.log
echo "bench rc=$?"; grep -E "timed steps|parity|secondary:|ceilings|phase split|families" $O/r5_bench_default_2.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_default_2.json"))
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 1), "prefill", round(d["prefill_ms"]), "decode", round(d["decode_ms"]))
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "frac_of_sustained")})
for s in d.get("secondary", []):
    print("sec", s["workload"][:44], round(s["value"]), s["ms_each_step"], s.get("parity_in_run", {}).get("ok"))
PY
