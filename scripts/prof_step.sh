#!/bin/bash
# whole-step kernel trace (run on the GPU box); summaries are copied to profiles/ by hand afterwards
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py --steps 10 --warmup 3 > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
tail -1 gpurun_out/prof/bench.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/step -o step --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/step.log 2>&1
ls gpurun_out/prof/step
