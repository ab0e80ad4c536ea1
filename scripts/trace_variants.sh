#!/bin/bash
# tracer kernel variants (compile-time switches of nero_amd/csrc/bvh.hip) through scripts/trace_bench.py, on the GPU box;
# VARIANTS: one set of flags per line ("-" = none)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
OBJS=$(ls build/obj/*.o | grep -v bvh.o | tr '\n' ' ')
export TRACE_CASES=${TRACE_CASES:-0,1} TRACE_MODES=0,1
LIST=${VARIANTS:-$'-\n-DPL_MIN_BLOCKS=6\n-DPL_BOTH_LEAVES\n-DPL_THREADS_X=128\n-DPL_THREADS_X=64'}
while read -r V; do
  [ "$V" = "-" ] && V=""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $V -c nero_amd/csrc/bvh.hip -o /tmp/bvh_var.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libvar.so $OBJS /tmp/bvh_var.o
  echo "== variant [$V]"
  NERO_HIP_LIB=/tmp/libvar.so timeout 200 python scripts/trace_bench.py 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('  ', d['case'], {k: v for k, v in d.items() if k.startswith('ms_')}, {k: (v['depth_and_position_bit_identical'], v['normals_bit_identical'], v['speedup']) for k, v in d.items() if '_vs_' in k})
"
done <<< "$LIST"
