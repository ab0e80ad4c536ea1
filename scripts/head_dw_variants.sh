#!/bin/bash
# head weight-gradient kernel variants (compile-time switches of nero_amd/csrc/mlp_engine.hip) through scripts/bench_head_dw.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OBJS=$(ls build/obj/*.o | grep -v mlp_engine.o | tr '\n' ' ')
LIST=${VARIANTS:-$'-\n-DHD_BALANCED=512\n-DHD_BALANCED=384\n-DHD_BALANCED=256'}
while read -r V; do
  [ "$V" = "-" ] && V=""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $V -c nero_amd/csrc/mlp_engine.hip -o /tmp/eng_var.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvar.so $OBJS /tmp/eng_var.o
  echo "== variant [$V]"
  NERO_HIP_LIB=/tmp/libvar.so timeout 100 python scripts/bench_head_dw.py 2>&1 | grep "us "
done <<< "$LIST"
echo "== library of the previous commit (git stash not available here: see the numbers in DESIGN.md)"
