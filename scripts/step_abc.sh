#!/bin/bash
# same-box comparison of several builds of libnero_hip.so on the training step: scripts/step_abc.sh lib1.so lib2.so ... (two rounds)
cd "$(dirname "$0")/.."
run() { echo "== $1"; env NERO_HIP_LIB=$PWD/$1 timeout 250 python bench.py --quick --steps 12 --warmup 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: v['ms_per_step'] for k, v in d.get('roofline',{}).get('per_kernel',{}).items()})"; }
for r in 1 2; do for l in "$@"; do run $l; done; done
