#!/bin/bash
# timing variants of the fp16 chain kernels (compile-time switches in mlp_f16_util.h / mlp_f16x3.hip), all with the
# per-phase shader-clock instrumentation.  Build here, then on the GPU box:
#   for v in build/variants/libv_*.so; do NERO_HIP_LIB=$PWD/$v python scripts/phase_timing.py 524288 f16x3 quick; done
# usage: scripts/f16_variants.sh name1:"-DFLAG ..." name2:"..." ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
OBJS=$(ls build/obj/*.o | grep -v "mlp_f16x3.o")
build() {
  local name=$1 flags=$2
  for f in mlp_f16x3; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -DF16_PHASE_TIMING $flags -c nero_amd/csrc/$f.hip -o build/variants/${f}_$name.o 2>/dev/null
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libv_$name.so $OBJS build/variants/mlp_f16x3_$name.o
}
for spec in "$@"; do build "${spec%%:*}" "${spec#*:}" & done
wait
ls build/variants/libv_*.so
