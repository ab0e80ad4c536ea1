#!/bin/bash
# round-3 GPU call 1: hardware probes, host facts, the whole GPU test tier with durations, a quick same-box bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(nproc; free -g | head -2; python -c "import torch;print(torch.cuda.get_device_name(0), torch.cuda.mem_get_info())") > gpurun_out/host.txt 2>&1
bash scripts/probe/run_fuse_probe.sh
timeout 1000 python -m pytest tests -m gpu -q --durations=40 -p no:cacheprovider > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc $?"
tail -60 gpurun_out/pytest_gpu.txt
timeout 300 python bench.py --quick --steps 10 --warmup 3 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; tail -1 gpurun_out/bench_quick.json | cut -c1-1500
timeout 300 python scripts/bench_material_step.py 4096 128 128 7 bell > gpurun_out/bench_material_step.txt 2>&1; grep fused gpurun_out/bench_material_step.txt
