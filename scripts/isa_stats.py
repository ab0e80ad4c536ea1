"""compile one .hip for gfx950 into build/isa/<tag>/ and print every kernel's register / spill / scratch figures (no GPU needed).
usage: python scripts/isa_stats.py nero_amd/csrc/mlp_f16x3.hip [tag] [-DFLAG ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, sys.argv[1]) if not os.path.isabs(sys.argv[1]) else sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else 'base'
flags = [a for a in sys.argv[2:] if a.startswith('-')]
out = os.path.join(ROOT, 'build', 'isa', tag)
os.makedirs(out, exist_ok=True)
base = os.path.basename(src)[:-4]
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed'] + flags +
                      ['-c', src, '-o', os.path.join(out, base + '.o'), '--save-temps=obj'], cwd=out)
s = open(os.path.join(out, base + '-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
for blk in re.split(r'\n  - \.agpr_count', s)[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r'\(anonymous namespace\)::|^void ', '', dem).split('(')[0][-60:]
    print(f"{short:62s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s}")
