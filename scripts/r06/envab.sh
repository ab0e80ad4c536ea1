#!/bin/bash
# same-box comparison of environment settings on the in-tree build: scripts/r06/envab.sh <out-name> "ENV=V ENV=V" "..." ...   (BENCH_ARGS, STEPS, REPS)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06
OUT=gpurun_out/r06/$1.txt; shift
: > $OUT
for rep in $(seq 1 ${REPS:-2}); do for envs in "$@"; do
  echo "== $envs" | tee -a $OUT
  env $envs timeout 300 python bench.py --quick --steps ${STEPS:-16} --warmup 4 ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k.replace('_kernel','').replace('_f16',''): v['ms_per_step'] for k, v in d.get('roofline',{}).get('per_kernel',{}).items()})
except Exception as e: print('FAILED', e)" | tee -a $OUT
done; done
