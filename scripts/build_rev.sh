#!/bin/bash
# build libnero_hip.so of a git revision (sources under nero_amd/csrc + include/) into build/variants/lib_<name>.so -- the baseline
# of same-box A/B runs (NERO_HIP_LIB=... selects it).  usage: scripts/build_rev.sh <name> <rev> [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
name=$1; rev=$2; shift 2
T=build/rev_$name
rm -rf $T; mkdir -p $T/nero_amd build/variants
git archive $rev nero_amd/csrc include | tar -x -C $T
objs=""
for f in $T/nero_amd/csrc/*.hip; do
  o=$T/$(basename ${f%.hip}).o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed "$@" -c $f -o $o &
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$name.so $objs
ls -la build/variants/lib_$name.so
