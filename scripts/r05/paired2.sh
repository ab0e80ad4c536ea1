#!/bin/bash
# NERO_F16_PAIRED masks at the other two workloads: 512 rays (configs[0]) and the Stage-II step
cd "$(dirname "$0")/../.."
BENCH_ARGS="--rays 512" REPS=2 STEPS=40 scripts/r05/envab.sh paired_ab_512 "NERO_F16_PAIRED=0" "NERO_F16_PAIRED=1" "NERO_F16_PAIRED=3" "NERO_F16_PAIRED=7"
BENCH_ARGS="--stage 2" REPS=2 STEPS=30 scripts/r05/envab.sh paired_ab_stage2 "NERO_F16_PAIRED=0" "NERO_F16_PAIRED=1" "NERO_F16_PAIRED=4" "NERO_F16_PAIRED=5"
REPS=2 STEPS=16 scripts/r05/envab.sh paired_ab_4096b "NERO_F16_PAIRED=0" "NERO_F16_PAIRED=3"
