#!/bin/bash
# build timing variants of the row-owner engine (compile-time switches in mlp_ro.hip) next to the product library; run on the GPU box:
#   for v in base nowait nomfma noepi nodma; do NERO_HIP_LIB=build/variants/libnero_$v.so python scripts/bench_ro.py 524288 f16x3r; done
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
OBJS=$(ls build/obj/*.o | grep -v mlp_ro.o)
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed $2 -c nero_amd/csrc/mlp_ro.hip -o build/variants/ro_$1.o && hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libnero_$1.so $OBJS build/variants/ro_$1.o; }
build base "" & build nowait "-DRO_NO_WAIT" & build nomfma "-DRO_NO_MFMA" & build noepi "-DRO_NO_EPI" & build nodma "-DRO_NO_DMA" &
wait
ls -la build/variants/*.so
