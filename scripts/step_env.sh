#!/bin/bash
# the training step under several environment settings, same box, two rounds: scripts/step_env.sh "A=1" "B=2" ...
cd "$(dirname "$0")/.."
run() { echo "== $1"; env $1 timeout 250 python bench.py --quick --steps 12 --warmup 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: v['ms_per_step'] for k, v in d.get('roofline',{}).get('per_kernel',{}).items()})"; }
for r in 1 2; do for e in "$@"; do run "$e"; done; done
