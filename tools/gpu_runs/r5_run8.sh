#!/bin/bash
# round 5, GPU call 10: LayerNorm folded into the QKV / h->4h GEMMs (bt_rowstats + folded epilogue): kernel tests, model parity (fold on /
# off, full-depth tests on the folded default), one-setting-per-process end-to-end A/B on configs[1] and configs[2]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -s -m gpu -k "lnfold or rowstats or layernorm" > $O/r5_run8_kernel_tests.log 2>&1
RC=$?
echo "kernel tests rc=$RC"; grep -E "^\[lnfold|passed|failed|Error" $O/r5_run8_kernel_tests.log | tail -16
if [ $RC -ne 0 ]; then tail -30 $O/r5_run8_kernel_tests.log; exit 0; fi
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -s -m gpu -k "layernorm_folded or full_depth or graphs_follow_reloaded or checkpoint_roundtrip or real_dims" > $O/r5_run8_model_tests.log 2>&1
echo "model tests rc=$?"; grep -E "^\[|passed|failed|Error" $O/r5_run8_model_tests.log | tail -30
for v in 0 1 0 1; do
  timeout 300 python tools/bench_knobs.py ln_fold=$v -- --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-parity --no-ceilings > $O/r5_ab_lnfold${v}_p2048_$RANDOM.json 2> /dev/null
done
for v in 0 1; do
  timeout 400 python tools/bench_knobs.py ln_fold=$v -- --prefill 128 --decode 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings > $O/r5_ab_lnfold${v}_p128.json 2> /dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_ab_lnfold*_p2048_*.json")) + sorted(glob.glob("gpurun_out/r5_ab_lnfold*_p128.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms prefill", round(d.get("prefill_ms", 0)), "decode", round(d.get("decode_ms", 0)), d.get("kernel_family_ms", {}).get("layernorm"), d.get("kernel_family_ms", {}).get("linear_mfma"))
    except Exception as e:
        print(f, "ERR", e)
PY
