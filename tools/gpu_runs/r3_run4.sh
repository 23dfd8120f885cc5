#!/bin/bash
# round 3, GPU call 4: pieces of the persistent kernel's per-tile epilogue + default bench line with the new K-loop
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 300 python tools/kernel_probe.py ppepi > $O/r3_ppepi.jsonl 2> $O/r3_ppepi.err
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r3_ppepi.jsonl"):
    r = json.loads(l)
    d[(r["M"], r["N"], r["K"], r["what"])].append(r["us"])
for k, v in d.items():
    print(k, v)
PY
tail -2 $O/r3_ppepi.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r3_bench_v2.json 2> $O/r3_bench_v2.err
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench_v2.json').read().strip().splitlines()[-1])
print('tok/s',round(d['value']),'ms/step',round(d['ms_per_step'],1),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1))
print(d['kernel_family_ms']); print(d['roofline']['frac'])"
tail -3 $O/r3_bench_v2.err
