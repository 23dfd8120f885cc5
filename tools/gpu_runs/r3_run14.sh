#!/bin/bash
# round 3, GPU call 14: CU-partitioned streams: attention of one half batch beside the GEMM chain of the other
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 200 python tools/cumask_probe.py --batch 2048 --rows 2048 > $O/r3_cumask_probe.jsonl 2> $O/r3_cumask_probe.err
cat $O/r3_cumask_probe.jsonl; tail -3 $O/r3_cumask_probe.err
timeout 200 python tools/cumask_probe.py --batch 2048 --rows 4096 >> $O/r3_cumask_probe.jsonl 2>> $O/r3_cumask_probe.err
tail -5 $O/r3_cumask_probe.jsonl
