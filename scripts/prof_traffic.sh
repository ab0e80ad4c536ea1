#!/bin/bash
# HBM traffic counters of the step's kernels (separate --pmc passes, as the microarch guide prescribes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof/fetch -o fetch --output-format csv -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline > gpurun_out/prof/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof/write -o write --output-format csv -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline > gpurun_out/prof/write.log 2>&1
ls gpurun_out/prof/fetch gpurun_out/prof/write
