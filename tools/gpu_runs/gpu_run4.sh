#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "ring_depth or mid_rows or mid_batch or linear_splitk or test_linear or pipeline or graphs_follow or measure_generation or batch_sizes_around or topk" ) > $O/r2_pytest4.log 2>&1
echo "pytest rc=$?" >> $O/r2_pytest4.log; tail -12 $O/r2_pytest4.log
for B in 64 256 32 128; do
  timeout 200 python bench.py --batch $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/r2_mid_b$B.json 2> $O/r2_mid_b$B.err
  python -c "import json;d=json.loads(open('$O/r2_mid_b$B.json').read().strip().splitlines()[-1]);print('B',$B,'tok/s',round(d['value']),'prefill',round(d['prefill_ms'],1),'decode',round(d['decode_ms'],1),'decode tok/s',round(d['decode_phase_tokens_per_s_per_gpu']))"
done
# ring depth A/B on the decode GEMM shapes (small-tile family): stages 2 vs default
python - <<'PY' > $O/r2_ring_ab.jsonl 2>$O/r2_ring_ab.err
import json, sys, os, torch
sys.path.insert(0, os.getcwd())
from block_transformer_amd import _lib, ops
_lib.use_debug_library(True)
lib=_lib.lib()
def timed(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b)/iters
for (M,N,K) in [(4096,2048,2048),(4096,6144,2048),(4096,2048,8192),(2048,2048,2048),(2048,6144,2048),(2048,8192,2048),(512,6144,2048),(512,2048,8192),(256,6144,2048),(256,2048,2048),(64,6144,2048),(64,2048,8192),(64,8192,2048)]:
    x=torch.randn(M,K,device="cuda").bfloat16(); w=(torch.randn(N,K,device="cuda")*K**-0.5).bfloat16(); b=torch.randn(N,device="cuda").bfloat16()
    out=torch.empty(M,N,device="cuda",dtype=torch.bfloat16)
    sc=torch.empty(16*M*N,dtype=torch.float32,device="cuda") if M<=256 else None
    for rep in range(2):
        for st in (2,0):
            lib.bt_debug_small_stages(st)
            ms=timed(lambda: ops.linear(x,w,b,out=out,scratch=sc))
            print(json.dumps({"op":"ring_ab","M":M,"N":N,"K":K,"stages":"2" if st==2 else "default(3/4)","us":round(ms*1e3,2),"TF":round(2*M*N*K/ms/1e9,1),"W_GBps":round(N*K*2/ms/1e6,1)}),flush=True)
lib.bt_debug_small_stages(0)
PY
cat $O/r2_ring_ab.jsonl | python -c "
import sys,json,collections
d=collections.defaultdict(list)
for l in sys.stdin:
    r=json.loads(l); d[(r['M'],r['N'],r['K'],r['stages'])].append(r['us'])
for k,v in d.items(): print(k, [round(x,1) for x in v])
"
tail -3 $O/r2_ring_ab.err
