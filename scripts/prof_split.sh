#!/bin/bash
# PMC passes over the chain / dW kernels of the default configuration (run on the GPU box); summary -> gpurun_out/prof/pmc_summary.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/prof/spmc1 -o spmc1 --output-format csv -- python scripts/dbg_split3.py 262144 > gpurun_out/prof/spmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-trace -d gpurun_out/prof/spmc2 -o spmc2 --output-format csv -- python scripts/dbg_split3.py 262144 > gpurun_out/prof/spmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for tag in ('spmc1', 'spmc2'):
    f = glob.glob(f'gpurun_out/prof/{tag}/**/*counter_collection.csv', recursive=True)
    if not f: print('no file', tag); continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if tag == 'spmc1' and r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[k] += 1
keep = [k for k in acc if any(s in k for s in ('_f16_kernel', '_split_kernel', 'mlp_', 'dw_gemm'))]
cols = sorted({c for k in keep for c in acc[k]})
with open('gpurun_out/prof/pmc_summary.csv', 'w') as fo:
    fo.write('# rocprofv3 --pmc (two passes) over scripts/dbg_split3.py 262144 (8x(256->256) chains in f32 / bf16x6 / f16x3 modes); sums over launches\n')
    fo.write('# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE)   (1024 SIMDs, counters accumulated per XCD)\n')
    fo.write('kernel,launches,mfma_busy,' + ','.join(cols) + '\n')
    for k in keep:
        d = acc[k]
        busy = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (128 * d['GRBM_GUI_ACTIVE']) if d.get('GRBM_GUI_ACTIVE') else 0
        fo.write(f'{k},{n[k]},{busy:.3f},' + ','.join(f'{d.get(c, 0):.4g}' for c in cols) + '\n')
print(open('gpurun_out/prof/pmc_summary.csv').read())
PY
