#!/bin/bash
# PMC pass over the split-bf16 forward kernel (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/prof/spmc1 -o spmc1 --output-format csv -- python scripts/dbg_split3.py > gpurun_out/prof/spmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-trace -d gpurun_out/prof/spmc2 -o spmc2 --output-format csv -- python scripts/dbg_split3.py > gpurun_out/prof/spmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ('spmc1', 'spmc2'):
    f = glob.glob(f'gpurun_out/prof/{tag}/**/*counter_collection.csv', recursive=True)
    if not f: print('no file', tag); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, d in acc.items():
        if 'split' in k or 'mlp_' in k:
            print(tag, k, {c: f'{v:.4g}' for c, v in d.items()})
PY
