#!/bin/bash
# round 4, GPU call 18 (final tree): the whole -m gpu suite the way the driver runs it (-x -q)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 ) > $O/r4_pytest_full_final2.log 2>&1
grep -v "^\.\|^$" $O/r4_pytest_full_final2.log | tail -24 > $O/r4_pytest_gpu_final_tail.txt; cat $O/r4_pytest_gpu_final_tail.txt
