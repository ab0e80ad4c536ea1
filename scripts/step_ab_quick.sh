#!/bin/bash
# same-box A/B of two library builds on the Stage-I step (scripts/step_times.py): scripts/step_ab_quick.sh libA.so libB.so [rays] [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for k in 1 2; do for L in $1 $2; do echo -n "$L: "; NERO_HIP_LIB=$R/$L timeout 200 python scripts/step_times.py ${3:-4096} ${4:-30} 2>&1 | tail -1; done; done
