#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in base solo; do echo "== $v"; NERO_HIP_LIB=$PWD/build/variants/libv_$v.so timeout 200 python scripts/phase_timing.py 524288 f16x3p quick 2>&1 | grep -v amdgpu | tee -a gpurun_out/phase_solo.txt; done
