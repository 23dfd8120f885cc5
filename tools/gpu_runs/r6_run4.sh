#!/bin/bash
# round 6, GPU call 4: (1) in-situ A/B of the tile boundary, one setting per process on one box (bt_debug_serial_epilogues = the round-5
# boundary), headline + P128/D2048; (2) timeline of the round-5 boundary (variant 95) for the record; (3) QKV one-batch RoPE loads on the new
# boundary; (4) the <= 4-row persistent-layer probe + launch gaps of the B = 1 / 4 block step from a rocprofv3 kernel trace;
# (5) the full GPU suite on the round-6 tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for v in 1 0 1 0; do
  timeout 300 python tools/bench_knobs.py serial_epilogues=$v -- --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-parity > $O/r6_ab_serial_epi${v}_p2048_$RANDOM.json 2> /dev/null
done
for v in 1 0; do
  timeout 400 python tools/bench_knobs.py serial_epilogues=$v -- --prefill 128 --decode 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity --no-ceilings > $O/r6_ab_serial_epi${v}_p128.json 2> /dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_ab_serial_epi*_p2048_*.json")) + sorted(glob.glob("gpurun_out/r6_ab_serial_epi*_p128.json")):
    try:
        d = json.load(open(f))
        c = d.get("ceilings") or {}
        print(f.split("/")[-1], round(d["value"]), "tok/s", round(d["ms_per_step"], 1), "ms prefill", round(d.get("prefill_ms", 0)), "decode", round(d.get("decode_ms", 0)),
              "gemm", d.get("kernel_family_ms", {}).get("linear_mfma"), "sclk", c.get("sclk_MHz_mean"), "W", c.get("power_W_mean"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 300 python tools/kernel_probe.py pptl --variant 95 > $O/r6_pp_timeline_round5_boundary.jsonl 2> /dev/null
timeout 300 python tools/kernel_probe.py pptl > $O/r6_pp_timeline_round6_boundary.jsonl 2> /dev/null
python - <<'PY'
import json
for f in ("round5", "round6"):
    for l in open(f"gpurun_out/r6_pp_timeline_{f}_boundary.jsonl"):
        try: d = json.loads(l)
        except Exception: continue
        if d.get('op') != 'pp_timeline' or d['M'] != 32768: continue
        print(d['boundary'], d['mode'], d['M'], d['N'], 'us', d['us_production'], d['us_with_stamps'], "7>1'", d['median_ticks']["7>1' last cluster issued -> next tile's first cluster opens"], 'tile', d['median_ticks']["1>1' whole tile"],
              'g0', list(d['group0_median_ticks'].values()), 'g1', list(d['group1_median_ticks'].values()))
PY
timeout 300 python tools/kernel_probe.py ppqkv --variant 12 --ab-other 29 > $O/r6_ppqkv_rope_one_batch_new_boundary.jsonl 2>/dev/null
cut -c1-300 $O/r6_ppqkv_rope_one_batch_new_boundary.jsonl
timeout 300 python tools/persistent_layer_probe.py > $O/r6_probe_persistent_layer.jsonl 2> $O/r6_probe_persistent_layer.err
echo "persistent probe rc=$?"; tail -3 $O/r6_probe_persistent_layer.err; cat $O/r6_probe_persistent_layer.jsonl
for B in 1 4; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r6_prof_b$B" -- python "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-secondary --no-ceilings ) > $O/r6_prof_b$B.log 2>&1
  python tools/summarize_prof.py gaps $O/r6_prof_b$B/*/ > $O/r6_launch_gaps_b$B.json 2>&1
  python tools/summarize_prof.py stats $O/r6_prof_b$B/*/ --title "rocprofv3 --kernel-trace --stats: bench.py --batch $B --steps 3 --warmup 1 (P2048/D128), round 6" --top 12 > $O/r6_bench_b${B}_kernel_stats.md 2>&1
  rm -rf $O/r6_prof_b$B
  cat $O/r6_launch_gaps_b$B.json
done
timeout 1500 python -m pytest tests -q -m gpu -x > $O/r6_run4_gpu_suite.log 2>&1
echo "gpu suite rc=$?"; tail -5 $O/r6_run4_gpu_suite.log
