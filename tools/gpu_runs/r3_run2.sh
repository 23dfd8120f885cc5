#!/bin/bash
# round 3, GPU call 2: what the persistent GEMM's epilogue costs and why (store burst vs instruction overhead)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 300 python tools/kernel_probe.py ppstore > $O/r3_ppstore.jsonl 2> $O/r3_ppstore.err
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r3_ppstore.jsonl"):
    r = json.loads(l)
    if r["op"] == "pp_store_path":
        d[(r["M"], r["N"], r["K"], r["case"])].append(r["us"])
    else:
        print(r)
for k, v in d.items():
    print(k, v)
PY
tail -3 $O/r3_ppstore.err
