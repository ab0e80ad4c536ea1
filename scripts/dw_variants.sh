#!/bin/bash
# timing variants of the fp16 weight-gradient kernel: scripts/dw_variants.sh name:"-Dflags" ...; then on the GPU box
#   for v in build/variants/libdw_*.so; do NERO_HIP_LIB=$PWD/$v python scripts/bench_dw.py 524288 quick; done
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
OBJS=$(ls build/obj/*.o | grep -v "mlp_f16dw.o")
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed $2 -c nero_amd/csrc/mlp_f16dw.hip -o build/variants/dw_$1.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libdw_$1.so $OBJS build/variants/dw_$1.o; }
for spec in "$@"; do build "${spec%%:*}" "${spec#*:}" & done
wait
ls build/variants/libdw_*.so
