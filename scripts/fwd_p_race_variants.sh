#!/bin/bash
# which ingredient of fwd_p_kernel's sporadic single-column corruption?  rebuild mlp_f16p.o with a switch, relink, replay.
# (the switches that needed source edits -- listed in DESIGN.md -- were removed again with the experiment code)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
cp nero_amd/libnero_hip.so /tmp/lib_orig.so
OBJS=$(ls build/obj/*.o | grep -v mlp_f16p.o | tr '\n' ' ')
for V in ${VARIANTS:-"" "-DNERO_PLAIN_STORES" "-DP_LDS_EXTRA=2560" "-DP_LDS_EXTRA=4096"}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $V -c nero_amd/csrc/mlp_f16p.hip -o /tmp/p_var.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o nero_amd/libnero_hip.so $OBJS /tmp/p_var.o
  echo "== variant [$V]"
  python scripts/replay_fwd_chain.py 2>&1 | grep "launches differing\|Error\|error" | cut -c1-400
done
cp /tmp/lib_orig.so nero_amd/libnero_hip.so
