#!/bin/bash
# round 3, GPU call 1: new parity tests (benchmark-batch dispatch, configs[3], product-library QKV shapes, margin-aware gates),
# timing ablations of the production persistent GEMM, decode-shape dispatch, vendor yardstick
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -s \
   -k "benchmark_batch or p2048_d2048 or long_prompt_long_decode or product_library or tracks_fp32 or streamer_stack or mid_batch_generate" ) > $O/r3_pytest1.log 2>&1
echo "pytest rc=$?" >> $O/r3_pytest1.log
grep -E "passed|failed|error|margin-aware|B=4096|rc=" $O/r3_pytest1.log | tail -40
timeout 300 python tools/kernel_probe.py ppabl > $O/r3_ppabl.jsonl 2> $O/r3_ppabl.err
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r3_ppabl.jsonl"):
    r = json.loads(l)
    if r["op"] == "pp_ablation":
        d[(r["M"], r["N"], r["K"], r["variant"], r["what"])].append(r["us"])
    else:
        print(r)
for k, v in d.items():
    print(k, v)
PY
tail -3 $O/r3_ppabl.err
timeout 300 python tools/kernel_probe.py decode > $O/r3_decode_shapes.jsonl 2> $O/r3_decode.err
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r3_decode_shapes.jsonl"):
    r = json.loads(l)
    d[(r["M"], r["N"], r["K"], r["variant"])].append(round(r["ms"] * 1e3, 1))
for k, v in d.items():
    print(k, v)
PY
timeout 300 python tools/kernel_probe.py vendor > $O/r3_vendor.jsonl 2> $O/r3_vendor.err
cat $O/r3_vendor.jsonl | cut -c1-200
