#!/usr/bin/env python
"""Summarise rocprofv3 output (ROCm 7: rocpd sqlite `*_results.db`) into the markdown / json kept under profiles/.

    # per-kernel totals of a `rocprofv3 --kernel-trace --stats -d DIR -o NAME -- <cmd>` run
    python tools/summarize_prof.py stats gpurun_out/prof_stats3 --title "bench.py, B=4096" > profiles/xyz_kernel_stats.md

    # one `--pmc COUNTER...` pass: average counter values per launch of the kernels matching a pattern
    python tools/summarize_prof.py pmc gpurun_out/prof3_gemm_FETCH_SIZE --kernel gemm_bf16

    # HBM traffic record for profiles/pmc_traffic.json from two separate passes (gfx950: FETCH_SIZE counts half of a wide
    # coalesced streaming read, so bytes = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB -- MI355X_MICROARCH.md, HBM section)
    python tools/summarize_prof.py traffic DIR_FETCH DIR_WRITE --kernel gemm_bf16 --shape "M=32768 N=6144 K=2048" \
        --algorithmic-bytes 562036736 --key linear_mfma --update profiles/pmc_traffic.json
"""
import argparse
import glob
import json
import os
import sqlite3
import sys


def _db(path):
    if os.path.isdir(path):
        cands = sorted(glob.glob(os.path.join(path, "*.db")))
        if not cands:
            sys.exit(f"no *.db under {path}")
        path = cands[0]
    return sqlite3.connect(path).cursor()


def _short(name, width=100):
    return name.replace("void ", "").replace("bt::", "")[:width]


def stats(args):
    cur = _db(args.path)
    tot, n, span = cur.execute("select sum(duration), count(*), max(end) - min(start) from kernels").fetchone()
    print(f"# rocprofv3 --kernel-trace --stats: {args.title}\n")
    print(f"Total kernel time {tot / 1e6:.1f} ms in {n} launches over a {span / 1e6:.1f} ms span "
          f"(rocpd sqlite `kernels` view, durations in ns).\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for name, c, s, a, mn, mx in cur.execute(
            "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
            "group by name order by 3 desc limit ?", (args.top,)):
        print(f"| `{_short(name)}` | {c} | {s / 1e6:.2f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * s / tot:.2f} |")
    if args.by_grid:
        print(f"\n## kernels matching `{args.by_grid}` by workgroup count\n\n| kernel | workgroups | calls | total ms | avg us |\n"
              "|---|---|---|---|---|")
        for name, g, c, s, a in cur.execute(
                "select substr(name, 1, 70), grid_x / workgroup_x, count(*), sum(duration) / 1e6, avg(duration) / 1e3 "
                "from kernels where name like ? group by 1, 2 order by 4 desc limit 16", (f"%{args.by_grid}%",)):
            print(f"| `{_short(name, 70)}` | {g} | {c} | {s:.2f} | {a:.2f} |")


def gaps(args):
    """Idle time between consecutive kernels of a trace (end of one -> start of the next, same queue order): what the dependent launch
    boundaries inside the captured block step cost.  Gaps above --max-gap-us are host-side (graph launch, sync) and are reported apart."""
    cur = _db(args.path)
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    if args.skip_first:
        rows = rows[int(len(rows) * args.skip_first):]
    g, by_pred = [], {}
    host = 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        d = (s1 - e0) / 1e3
        if d > args.max_gap_us:
            host += 1
            continue
        g.append(d)
        by_pred.setdefault(_short(n0, 60), []).append(d)
    g.sort()
    tot_k = sum(e - s for _, s, e in rows) / 1e3
    out = {"kernels": len(rows), "device_side_boundaries": len(g), "host_side_gaps": host, "sum_kernel_us": round(tot_k, 1),
           "sum_gap_us": round(sum(g), 1), "gap_share_of_busy_span": round(sum(g) / (tot_k + sum(g)), 4),
           "gap_us": {"median": round(g[len(g) // 2], 3), "mean": round(sum(g) / len(g), 3), "p10": round(g[len(g) // 10], 3),
                      "p90": round(g[len(g) * 9 // 10], 3)},
           "median_gap_us_after_kernel": {k: round(sorted(v)[len(v) // 2], 3) for k, v in sorted(by_pred.items(), key=lambda kv: -len(kv[1]))[:10]}}
    print(json.dumps(out, indent=1))


def _pmc(path, kernel):
    cur = _db(path)
    rows = cur.execute("select counter_name, avg(value), avg(duration), count(*) from counters_collection "
                       "where kernel_name like ? group by counter_name", (f"%{kernel}%",)).fetchall()
    if not rows:
        sys.exit(f"no counters for kernels like {kernel!r} in {path}")
    return {r[0]: r[1] for r in rows}, rows[0][2] / 1e3, rows[0][3]


def pmc(args):
    vals, dur_us, n = _pmc(args.path, args.kernel)
    out = {"kernel_like": args.kernel, "launches": n, "avg_duration_us": dur_us, "counters": vals}
    if "SQ_WAVE_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
        wc, cyc = vals["SQ_WAVE_CYCLES"], vals["GRBM_GUI_ACTIVE"] / 8          # 8 XCDs
        out["derived"] = {
            "clock_GHz": cyc / (dur_us * 1e3),
            "wait_any_frac": vals.get("SQ_WAIT_ANY", 0) / wc,
            "wait_inst_frac": vals.get("SQ_WAIT_INST_ANY", 0) / wc,
            "active_inst_frac": vals.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            "mfma_busy_frac_of_simd_cycles": vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024),   # 256 CUs x 4 SIMDs
        }
    print(json.dumps(out, indent=1))


def traffic(args):
    f, dur, _ = _pmc(args.fetch_dir, args.kernel)
    w, _, _ = _pmc(args.write_dir, args.kernel)
    rec = {"shape": args.shape, "fetch_size_KiB_raw": f["FETCH_SIZE"], "write_size_KiB": w["WRITE_SIZE"],
           "hbm_bytes_corrected": 2 * f["FETCH_SIZE"] * 1024 + w["WRITE_SIZE"] * 1024,
           "algorithmic_bytes": args.algorithmic_bytes, "avg_duration_us": dur}
    if args.kernel_name:
        rec["kernel"] = args.kernel_name
    if args.git:
        rec["git"] = args.git
    if args.update:
        data = json.load(open(args.update)) if os.path.exists(args.update) else {}
        data[args.key] = rec
        json.dump(data, open(args.update, "w"), indent=1)
    print(json.dumps({args.key: rec}, indent=1))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("stats")
    a.add_argument("path")
    a.add_argument("--title", default="")
    a.add_argument("--top", type=int, default=14)
    a.add_argument("--by-grid", default="gemm_bf16")
    a.set_defaults(fn=stats)
    a = sub.add_parser("gaps")
    a.add_argument("path")
    a.add_argument("--max-gap-us", type=float, default=40.0)
    a.add_argument("--skip-first", type=float, default=0.5, help="fraction of the trace to skip (warm-up, prefill of the first generate)")
    a.set_defaults(fn=gaps)
    a = sub.add_parser("pmc")
    a.add_argument("path")
    a.add_argument("--kernel", required=True)
    a.set_defaults(fn=pmc)
    a = sub.add_parser("traffic")
    a.add_argument("fetch_dir")
    a.add_argument("write_dir")
    a.add_argument("--kernel", required=True)
    a.add_argument("--shape", required=True)
    a.add_argument("--algorithmic-bytes", type=int, required=True)
    a.add_argument("--key", required=True)
    a.add_argument("--update")
    a.add_argument("--kernel-name", default="", help="recorded in the json: the kernel the pass measured")
    a.add_argument("--git", default="", help="recorded in the json: revision of the tree the pass ran on")
    a.set_defaults(fn=traffic)
    args = ap.parse_args()
    args.fn(args)


if __name__ == "__main__":
    main()
