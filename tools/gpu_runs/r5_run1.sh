#!/bin/bash
# round 5, GPU call 1: the new parity tests (full-depth fp32 1e-3, full-size b8, ratcheted bf16 budget, streamer protocol) and the
# default bench line with in-run ceilings, clock / power sampling, last-block parity and checked secondary legs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python - > $O/r5_amdsmi_probe.txt 2>&1 <<'PY'
import json
try:
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    for name, fn in (("gpu_metrics", lambda: amdsmi.amdsmi_get_gpu_metrics_info(h)),
                     ("clock_gfx", lambda: amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)),
                     ("power", lambda: amdsmi.amdsmi_get_power_info(h))):
        try:
            v = fn()
            print(name, json.dumps({k: (str(x) if not isinstance(x, (int, float, list, str)) else x) for k, x in v.items()})[:1500])
        except Exception as e:
            print(name, "ERR", type(e).__name__, e)
except Exception as e:
    print("amdsmi ERR", type(e).__name__, e)
import glob
print(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/*")[:40])
PY
cat $O/r5_amdsmi_probe.txt | cut -c1-600
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -s -k "full_depth or compaction_under_sampling_and_streamer or streamer_protocol" > $O/r5_run1_tests.log 2>&1
echo "tests rc=$?"; grep -E "^\[|passed|failed|Error|error" $O/r5_run1_tests.log | tail -40
timeout 900 python bench.py --steps 3 --warmup 1 > $O/r5_bench_default_1.json 2> $O/r5_bench_default_1.log
echo "bench rc=$?"; tail -25 $O/r5_bench_default_1.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_default_1.json"))
print("value", d["value"], "ms", d["ms_per_step"])
print("ceilings", json.dumps(d.get("ceilings"))[:1500])
print("roofline", json.dumps(d.get("roofline"))[:800])
print("parity", json.dumps(d.get("parity_in_run"))[:900])
for s in d.get("secondary", []):
    print("sec", s["workload"][:40], s["value"], s["ms_each_step"], json.dumps(s.get("parity_in_run"))[:600])
PY
