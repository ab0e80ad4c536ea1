#!/bin/bash
# timing variants of the 64-row-tile fp16 engine (compile-time switches in mlp_f16x3.hip); run on the GPU box:
#   for v in base nomfma nowstream noepi; do NERO_HIP_LIB=$PWD/build/variants/libf16_$v.so python scripts/bench_ro.py 524288 f16x3; done
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
OBJS=$(ls build/obj/*.o | grep -v mlp_f16x3.o)
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed $2 -c nero_amd/csrc/mlp_f16x3.hip -o build/variants/f16_$1.o && hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libf16_$1.so $OBJS build/variants/f16_$1.o; }
build base "" & build nomfma "-DF16_NO_MFMA" & build nowstream "-DF16_NO_WSTREAM" & build noepi "-DF16_NO_EPI" & build nomfma_now "-DF16_NO_MFMA -DF16_NO_WSTREAM" &
wait
ls build/variants/libf16_*.so
