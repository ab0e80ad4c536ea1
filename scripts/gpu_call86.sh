#!/bin/bash
R=${GRAFT_REPO_ROOT}; cd $R && mkdir -p gpurun_out
bash scripts/gpu_trace_bench.sh
timeout 600 python -m pytest tests/test_tracer.py tests/test_material_render.py -x -q -m gpu 2>&1 | tail -4
