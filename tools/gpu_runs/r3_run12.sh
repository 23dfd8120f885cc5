#!/bin/bash
# round 3, GPU call 12: rocprofv3 kernel trace of the default bench (where the GEMM time is, by kernel and grid)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/r3_prof_b4096" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-roofline --no-cpu-baseline ) > $O/r3_prof_b4096.log 2>&1
python tools/summarize_prof.py stats $O/r3_prof_b4096/*/ --title "bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline (B = 4096, P2048/D128; 2 generate() calls: 1 timed + 1 phase split), round 3 mid-round tree" --top 20 > $O/r3_bench_b4096_kernel_stats_mid.md 2>&1
cat $O/r3_bench_b4096_kernel_stats_mid.md
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/r3_prof_b4096/*/*.db")[0]
cur = sqlite3.connect(db).cursor()
print("\n## persistent / small-tile GEMM launches by (kernel, grid, duration bucket)")
for row in cur.execute("select substr(name,1,60), grid_x/workgroup_x, count(*), round(sum(duration)/1e6,1), round(avg(duration)/1e3,1), round(min(duration)/1e3,1), round(max(duration)/1e3,1) from kernels where name like '%gemm_bf16%' group by 1,2, cast(duration/1e5 as int) order by 4 desc limit 40"):
    print(row)
PY
rm -rf $O/r3_prof_b4096
