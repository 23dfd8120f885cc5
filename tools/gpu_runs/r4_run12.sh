#!/bin/bash
# round 4, GPU call 12: the whole -m gpu suite with durations (what the driver runs at round end)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 1500 python -m pytest tests/ -q -m gpu --durations=40 ) > $O/r4_pytest_full.log 2>&1
grep -v "^\.\|^$" $O/r4_pytest_full.log | tail -60
