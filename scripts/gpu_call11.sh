#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_engines_extra.py tests/test_mlp_engine.py tests/test_stage1_driver.py tests/test_trainer_fusion.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
python scripts/prof_launches.py 4096 > gpurun_out/prof_launches_narrow.txt 2>&1; grep -E " dw |^\{" gpurun_out/prof_launches_narrow.txt | awk '$5 < 40 || /^\{/' 
timeout 200 python scripts/step_times.py 4096 30 2>&1 | grep driver=
