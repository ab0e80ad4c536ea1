#!/bin/bash
# same-box A/B of two builds of libnero_hip.so on the training step: scripts/step_ab.sh libA.so libB.so [env...]
cd "$(dirname "$0")/.."
run() { echo "== $1"; env NERO_HIP_LIB=$1 $2 timeout 250 python bench.py --quick --steps 12 --warmup 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: v['ms_per_step'] for k, v in d.get('roofline',{}).get('per_kernel',{}).items()})"; }
run $PWD/$1 "$3"; run $PWD/$2 "$3"; run $PWD/$1 "$3"; run $PWD/$2 "$3"
