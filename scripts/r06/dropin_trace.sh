#!/bin/bash
# kernel trace of the drop-in training path at 512 rays -> gpurun_out/r06/dropin_trace.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r06 gpurun_out/prof
rm -rf gpurun_out/prof/dropin
NO_CPROFILE=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/dropin -- python scripts/r06/dropin_profile.py 512 > gpurun_out/r06/dropin_trace.log 2>&1
f=$(find gpurun_out/prof/dropin -name "*kernel_trace.csv" | head -1)
python scripts/r06/dropin_trace_analysis.py $f > gpurun_out/r06/dropin_trace${TAG}.txt 2>&1
tail -3 gpurun_out/r06/dropin_trace.log; cat gpurun_out/r06/dropin_trace${TAG}.txt
rm -rf gpurun_out/prof/dropin
