#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x -k "golden or graph_capture or eos" ) 2>&1 | tail -2
