#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
for B in 64 256; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r2_prof2_b$B" -- python "$GRAFT_REPO_ROOT/bench.py" --batch $B --steps 1 --warmup 1 --no-roofline --no-cpu-baseline ) > $O/r2_prof2_b$B.log 2>&1
  python tools/summarize_prof.py stats $O/r2_prof2_b$B/*/ --title "bench.py --batch $B (P2048/D128), mid-batch slab path, 3 generate() calls" --top 20 > $O/r2_bench_b${B}_kernel_stats_v2.md 2>&1
  head -30 $O/r2_bench_b${B}_kernel_stats_v2.md
  grep -o '"value": [0-9.]*\|"decode_ms": [0-9.]*' $O/r2_prof2_b$B.log | tr '\n' ' '; echo
  python - <<PY
import sqlite3, glob
db=glob.glob("$O/r2_prof2_b$B/*/*.db")[0]
c=sqlite3.connect(db)
# timeline of one decode block step: take the last 700 kernels, print gaps
rows=c.execute("select name,start,end from kernels order by start").fetchall()
rows=rows[-1500:-500]
gaps=[rows[i+1][1]-rows[i][2] for i in range(len(rows)-1)]
dur=[r[2]-r[1] for r in rows]
import statistics
print("tail window: kernels",len(rows),"sum dur us",sum(dur)/1e3,"sum gaps us",sum(gaps)/1e3,"median gap us",statistics.median(gaps)/1e3,"span us",(rows[-1][2]-rows[0][1])/1e3)
for r in rows[200:236]:
    print(round((r[2]-r[1])/1e3,1), r[0][:70])
PY
done
