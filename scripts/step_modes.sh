#!/bin/bash
# the training step under different engine selections (run on the GPU box): headline ms/step of bench.py --quick
cd "$(dirname "$0")/.."
run() { echo "== $1"; env $2 timeout 250 python bench.py --quick --steps 12 --warmup 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), {k: v['ms_per_step'] for k, v in d.get('roofline',{}).get('per_kernel',{}).items()})"; }
if [ -z "$SKIP_BASE" ]; then
run "all f16x3, dW bf16x6" "NERO_GEMM_DW=bf16x6"
run "dW f16x3" "NERO_GEMM_DW=f16x3"
fi
