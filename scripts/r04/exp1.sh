#!/bin/bash
# round 4, GPU call 1: persistent chain kernels -- correctness of the engine tests, then same-box step timings per configuration
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/exp1.txt
: > $OUT
( timeout 600 python -m pytest tests/test_mlp_engine.py tests/test_engines_extra.py tests/test_determinism.py -m gpu -x -q 2>&1 | tail -5 ) | tee -a $OUT
run() {   # label, env...
  local label=$1; shift
  echo "== $label" | tee -a $OUT
  env "$@" timeout 300 python bench.py --quick --steps 16 --warmup 4 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: v['ms_per_step'] for k, v in d.get('roofline',{}).get('per_kernel',{}).items()})
except Exception as e: print('FAILED', e)" | tee -a $OUT
}
B=$PWD/build/variants/lib_base.so
run "base" NERO_HIP_LIB=$B
run "new persist=7" NERO_F16_PERSIST=7
run "new persist=0" NERO_F16_PERSIST=0
run "new persist=1 (fwd)" NERO_F16_PERSIST=1
run "new persist=3 (fwd+bwd)" NERO_F16_PERSIST=3
run "base" NERO_HIP_LIB=$B
run "new persist=7" NERO_F16_PERSIST=7
run "new p7 dw batch all, total 1024 group 12" NERO_DW_BATCH_ROWS=100000000
run "new p7 dw batch all, total 512 group 4" NERO_DW_BATCH_ROWS=100000000 NERO_DW_BATCH_TOTAL=512 NERO_DW_BATCH_GROUP=4
run "new p7 dw batch all, total 512 group 2" NERO_DW_BATCH_ROWS=100000000 NERO_DW_BATCH_TOTAL=512 NERO_DW_BATCH_GROUP=2
run "new p7 dw batch all, total 256 group 2" NERO_DW_BATCH_ROWS=100000000 NERO_DW_BATCH_TOTAL=256 NERO_DW_BATCH_GROUP=2
run "new p7 dw batch all, total 768 group 3" NERO_DW_BATCH_ROWS=100000000 NERO_DW_BATCH_TOTAL=768 NERO_DW_BATCH_GROUP=3
