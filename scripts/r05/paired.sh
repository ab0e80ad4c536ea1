#!/bin/bash
# two-workgroups-per-CU chain kernels (mlp_f16p.hip): identity tests, then a same-box A/B of NERO_F16_PAIRED masks
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_paired_engine.py -x -q --tb=short 2>&1 | tail -15 | tee gpurun_out/r05/paired_tests.txt
REPS=2 STEPS=16 scripts/r05/envab.sh paired_ab_4096 "NERO_F16_PAIRED=0" "NERO_F16_PAIRED=1" "NERO_F16_PAIRED=2" "NERO_F16_PAIRED=4" "NERO_F16_PAIRED=7"
