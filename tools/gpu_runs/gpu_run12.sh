#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/r2_pytest12_full.log 2>&1
echo "pytest rc=$?" >> $O/r2_pytest12_full.log; tail -6 $O/r2_pytest12_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
