#!/bin/bash
# on the GPU box: phase timing of every build/variants/libph_*.so given by name (quick = relu only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; mkdir -p gpurun_out/r06
MODE=${PHMODE:-quick}
for n in "$@"; do
  echo "=== $n"
  NERO_HIP_LIB=$R/build/variants/libph_$n.so NERO_F16_PAIRED=${PAIRED:-0} python scripts/phase_timing.py 524288 f16x3 $MODE 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r06/ph_$(date +%H%M%S).txt
