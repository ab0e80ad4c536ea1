#!/bin/bash
# whole-step kernel trace of the DEFAULT bench command (run on the GPU box); the summaries are copied to profiles/ afterwards
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
tail -1 gpurun_out/prof/bench.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/step -o step --output-format csv -- python bench.py --quick > gpurun_out/prof/step.log 2>&1
tail -1 gpurun_out/prof/step.log | cut -c1-300
ls gpurun_out/prof/step
