#!/bin/bash
# wave-per-ray sampler / compositing kernels, column-quad encoders: GPU tests + same-box step timing + kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/prof
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc $?"
tail -15 gpurun_out/pytest_gpu.txt
for rays in 4096 512; do
  timeout 200 python scripts/step_times.py $rays 30 2>&1 | grep driver= | tee -a gpurun_out/step_times.txt
done
rm -rf gpurun_out/prof/step4096
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/step4096 -o step --output-format csv -- python scripts/step_times.py 4096 30 > gpurun_out/prof/step4096.log 2>&1
rm -rf gpurun_out/prof/*/*.db gpurun_out/prof/*/*kernel_trace.csv
