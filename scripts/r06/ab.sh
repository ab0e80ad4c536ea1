#!/bin/bash
# same-box comparison of library builds on the Stage-I step: scripts/r06/ab.sh <out-name> <lib-or-'-'>[:ENV=V,ENV=V] ...   ('-' = the in-tree build)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06
OUT=gpurun_out/r06/$1.txt; shift
: > $OUT
run() {
  local spec=$1 lib=${1%%:*} envs=""
  [[ "$spec" == *:* ]] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  [[ "$lib" != "-" ]] && envs="$envs NERO_HIP_LIB=$PWD/build/variants/lib_$lib.so"
  echo "== $spec" | tee -a $OUT
  env $envs timeout 300 python bench.py --quick --steps ${STEPS:-16} --warmup 4 ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k.replace('_kernel','').replace('_f16',''): v['ms_per_step'] for k, v in d.get('roofline',{}).get('per_kernel',{}).items()})
except Exception as e: print('FAILED', e)" | tee -a $OUT
}
for rep in 1 2; do for s in "$@"; do run $s; done; done
