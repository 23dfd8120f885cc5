#!/bin/bash
# round 3, GPU call 5: does the cost of the epilogue's stores scale with their number?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 300 python tools/kernel_probe.py ppepi --split 2 > $O/r3_ppepi2.jsonl 2> $O/r3_ppepi2.err
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r3_ppepi2.jsonl"):
    r = json.loads(l)
    d[(r["M"], r["N"], r["K"], r["what"])].append(r["us"])
for k, v in d.items():
    print(k, v)
PY
tail -2 $O/r3_ppepi2.err
