#!/bin/bash
# determinism of the Stage-I step per library build and join position:  scripts/r05/hunt3.sh "<lib ...>" "<join modes>" [runs]
# e.g. scripts/build_variant.sh slp_shade shade -fslp-vectorize; scripts/r05/hunt3.sh "build/variants/lib_slp_shade.so nero_amd/libnero_hip.so" "j l" 8
cd "$(dirname "$0")/../.."
for lib in $1; do for j in $2; do for i in $(seq 1 ${3:-6}); do
  echo "$lib join=$j: $(NERO_HIP_LIB=$PWD/$lib NERO_STREAMS=3 NERO_DW_JOIN=$j python scripts/r05/dbg_streams.py bear 512 2>&1 | grep -c identical) of 9 repeats identical"
done; done; done
