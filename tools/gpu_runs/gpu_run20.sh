#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/r2_pytest_gpu_full2.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r2_pytest_gpu_full2.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
