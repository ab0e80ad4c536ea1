#!/bin/bash
# build phase-timing variants of the chain kernels: scripts/r06/phv.sh name1:"-DFLAG ..." name2:"..."  -> build/variants/libph_<name>.so
# (both chain units rebuilt with -DF16_PHASE_TIMING + the flags; run on the GPU box with
#  NERO_HIP_LIB=$PWD/build/variants/libph_<name>.so python scripts/phase_timing.py 524288 f16x3)
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/variants
OBJS=$(ls build/obj/*.o | grep -v "mlp_f16x3.o\|mlp_f16p.o")
build() {
  local name=$1 flags=$2
  for f in mlp_f16x3 mlp_f16p; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-pass-failed -DF16_PHASE_TIMING $flags -c nero_amd/csrc/$f.hip -o build/variants/${f}_ph_$name.o 2>/dev/null
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libph_$name.so $OBJS build/variants/mlp_f16x3_ph_$name.o build/variants/mlp_f16p_ph_$name.o
  rm -f build/variants/mlp_f16x3_ph_$name.o build/variants/mlp_f16p_ph_$name.o
}
for spec in "$@"; do build "${spec%%:*}" "${spec#*:}" & done
wait
ls build/variants/libph_*.so
