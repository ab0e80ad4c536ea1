#!/bin/bash
# VGPRs / spills / scratch of every kernel of one translation unit:  scripts/kernel_regs.sh <unit, e.g. mlp_f16x3> ["<extra flags>"]
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-pass-failed $2 -Rpass-analysis=kernel-resource-usage -c nero_amd/csrc/$1.hip -o /tmp/kr_$$.o 2>&1 |
  python3 -c "
import re,sys,subprocess
cur=None;rows={}
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln) or re.search(r' Name: (\S+)',ln)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'(VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)',ln)
    if m and cur: rows[cur][m.group(1)]=int(m.group(2))
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()[:70]
    print(f'{name:70s} vgpr {v.get(\"VGPRs\",0):4d} agpr {v.get(\"AGPRs\",0):3d} vspill {v.get(\"VGPRs Spill\",0):3d} sspill {v.get(\"SGPRs Spill\",0):3d} scratch {v.get(\"ScratchSize [bytes/lane]\",0):4d}')
"
rm -f /tmp/kr_$$.o
