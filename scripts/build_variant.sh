#!/bin/bash
# relink libnero_hip.so with ONE translation unit rebuilt under extra flags -> build/variants/lib_<name>.so  (NERO_HIP_LIB selects it)
# usage: scripts/build_variant.sh <name> <unit, e.g. mlp_f16x3> "<flags>"     (run __graft_entry__.py first: build/obj must be current)
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; flags=$3
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-pass-failed $flags -c nero_amd/csrc/$unit.hip -o build/variants/${unit}_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$name.so $(ls build/obj/*.o | grep -v "/$unit.o") build/variants/${unit}_$name.o
rm -f build/variants/${unit}_$name.o
echo build/variants/lib_$name.so
