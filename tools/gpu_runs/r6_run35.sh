#!/bin/bash
# round 6, GPU call 35: final tree -- smoke(), the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $O/r6_smoke_final.log 2>&1
echo "smoke rc=$?"; tail -2 $O/r6_smoke_final.log | cut -c1-200
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r6_gpu_suite.log 2>&1
echo "GPU suite rc=$?"; tail -3 $O/r6_gpu_suite.log | cut -c1-200
