#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_material_train.py tests/test_stage2_driver.py tests/test_parity_at_size.py -m gpu -q -x -p no:cacheprovider -k "glue or reg_points or bit_for_bit or two_ranks or fused_material or c_driven or stage2 or c4 or c5" 2>&1 | tail -8
for g in torch hip; do
  echo "== NERO_LOSS_GLUE=$g"
  NERO_LOSS_GLUE=$g python scripts/bench_material_step.py 4096 128 128 7 bell fused | tail -1
  NERO_LOSS_GLUE=$g python scripts/bench_material_step.py 2048 256 256 7 bear fused | tail -1
done
