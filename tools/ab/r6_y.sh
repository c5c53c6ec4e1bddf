#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06 B > gpurun_out/collect.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh r06 B 1 > gpurun_out/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_mstamps.so python tools/debug/merged_stamps.py B 1 gpurun_out/merged_timeline_r06.txt > gpurun_out/stamps.log 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r06.json; cut -c1-400 gpurun_out/bench_r06.json
