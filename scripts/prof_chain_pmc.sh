#!/bin/bash
# SQ / LDS counters of the chain forward kernels (separate rocprofv3 --pmc passes, kernel trace only), run on the GPU box:
#   bash scripts/prof_chain_pmc.sh [modes]   ->  gpurun_out/pmc/pass*/ , gpurun_out/pmc/summary.txt
cd "$(dirname "$0")/.." && ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*\|MfmaUtil\|VALUBusy" | sort -u > $OUT/avail.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM"
P4="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  ( cd /tmp && timeout 170 rocprofv3 --pmc $P --kernel-trace -d $OUT/pass$i -o p --output-format csv -- python $ROOT/scripts/pmc_chain.py $1 > $OUT/pass$i.log 2>&1 )
  tail -2 $OUT/pass$i.log
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get('OUT', 'gpurun_out/pmc')
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'fwd' not in k: continue
        name = 'fwd_f16'
        rows[name][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for name, d in rows.items():
        fo.write(name + '\n')
        for c, v in sorted(d.items()):
            fo.write(f'  {c:36s} {sum(v)/len(v):16.0f}  (n={len(v)})\n')
print(open(out + '/summary.txt').read())
PY
