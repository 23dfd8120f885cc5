#!/usr/bin/env python
"""Condense rocprofv3 output directories into small committed summaries (profiles/).

    python tools/summarize_prof.py stats  <rocprof_out_dir> <out.md>     # --kernel-trace --stats run
    python tools/summarize_prof.py pmc    <rocprof_out_dir> <out.json>   # --pmc <COUNTER> run (per-kernel averages)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def short(name):
    name = name.replace("void ", "")
    for pre in ("bt::",):
        name = name.replace(pre, "")
    return name[:110]


def stats(root, out):
    files = find(root, "*kernel_stats.csv")
    if not files:
        raise SystemExit(f"no *kernel_stats.csv under {root}")
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(out, "w") as o:
        o.write(f"rocprofv3 --kernel-trace --stats summary ({os.path.basename(files[0])}); total kernel time "
                f"{tot / 1e6:.2f} ms\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            o.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                    f"{float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | "
                    f"{float(r['Percentage']):.2f} |\n")
    print(open(out).read())


def pmc(root, out):
    files = find(root, "*counter_collection.csv")
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {root}")
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {c: {"avg": sum(v) / len(v), "n": len(v)} for c, v in d.items()} for k, d in acc.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
