for v in build/variants/libv_*.so; do for m in f16x3 f16x3p; do echo "== $v $m"; NERO_HIP_LIB=$PWD/$v timeout 100 python scripts/phase_timing.py 524288 $m quick 2>&1 | grep -v amdgpu.ids; done; done
