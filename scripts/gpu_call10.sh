#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in base sgb; do for m in f16x3 f16x3p; do echo "== $v $m"; NERO_HIP_LIB=$PWD/build/variants/libv_$v.so timeout 200 python scripts/phase_timing.py 524288 $m quick 2>&1 | grep -v amdgpu | tee -a gpurun_out/phase_sgb.txt; done; done
