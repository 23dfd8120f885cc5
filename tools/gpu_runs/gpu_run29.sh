#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x -k "full_depth or real_dims or golden or G2 or ragged or mask or eos or production_dims or properties" ) 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/r2_bench_default_final3.json 2> $O/r2_bench_default_final3.err
echo "bench default rc=$?"; python -c "import json;d=json.loads(open('$O/r2_bench_default_final3.json').read().strip().splitlines()[-1]);print(round(d['value'],1),d['steps'],d['warmup'],round(d['prefill_ms'],1),round(d['decode_ms'],1),d['roofline']['frac'],d['roofline_decode_phase']['frac_of_binding_roof'],d['kernel_family_ms'],d['cpu_baseline']['value'])"
