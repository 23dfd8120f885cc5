#!/bin/bash
# round 5, GPU call 9 (final tree): the default bench exactly as the driver runs it (--steps 20 --warmup 5), configs[3] per GPU, and
# the batch sweep for the results table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/r5_bench_default_final.json 2> $O/r5_bench_default_final.log ) 2> $O/r5_bench_default_final.time
echo "bench rc=$?"; cat $O/r5_bench_default_final.time | tail -4; grep -E "timed steps|parity|secondary:|ceilings|phase split" $O/r5_bench_default_final.log
timeout 400 python bench.py --prefill 2048 --decode 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings > $O/r5_bench_p2048_d2048.json 2> $O/r5_bench_p2048_d2048.log
for b in 1 2 4 8 16 32 64 128 256 1024; do
  timeout 120 python bench.py --batch $b --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings > $O/r5_sweep_b$b.json 2> $O/r5_sweep_b$b.log
done
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_default_final.json"))
print("default", round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms; prefill", round(d["prefill_ms"]), "decode", round(d["decode_ms"]))
print("ceilings", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["ceilings"].items() if k in ("copy_GBs", "mfma_sustained_TFLOPs", "sclk_MHz_mean", "power_W_mean", "mfma_loop_sclk_MHz_mean", "mfma_loop_power_W_mean")})
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "frac_of_sustained")}, "families", d["kernel_family_ms"])
print("decode roof", d["roofline_decode_phase"]["frac_of_binding_roof"], d["roofline_decode_phase"]["empirical"])
for s in d.get("secondary", []):
    print("sec", s["workload"][:44], round(s["value"]), s["ms_each_step"], s.get("parity_in_run", {}).get("ok"), s["roofline_decode_phase"]["frac_of_binding_roof"])
d = json.load(open("gpurun_out/r5_bench_p2048_d2048.json")); print("p2048/d2048", d["config"]["batch_per_gpu"], round(d["value"]))
for b in (1, 2, 4, 8, 16, 32, 64, 128, 256, 1024):
    d = json.load(open(f"gpurun_out/r5_sweep_b{b}.json")); print("B", b, round(d["value"], 1), round(d["ms_per_step"], 2))
PY
