#!/bin/bash
# PMC pass over the chain kernels (run on the GPU box): MFMA busy vs wave cycles / stalls.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/prof/pmc1 -o pmc1 --output-format csv -- python scripts/bench_engine.py 131072 > gpurun_out/prof/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d gpurun_out/prof/pmc2 -o pmc2 --output-format csv -- python scripts/bench_engine.py 131072 > gpurun_out/prof/pmc2.log 2>&1
ls -R gpurun_out/prof | head -30
