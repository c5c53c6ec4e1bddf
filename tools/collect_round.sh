#!/bin/bash
# Run ON THE GPU BOX (via gpurun), at the end of a round: everything that is copied into profiles/ afterwards.
#   tools/ablate.sh render_fwd_merged.hip mstamps=-DVAMP_MERGED_STAMPS      (here, before the call: the stamps build)
#   gpurun --timeout 4000 -- 'bash tools/collect_round.sh r06'
# Outputs under gpurun_out/: profiles_<tag>/ (tools/collect_profiles.sh), pmc_<tag>/summary.txt (tools/pmc_passes.sh),
# merged_timeline_<tag>.txt, pytest_gpu.log (the whole -m gpu suite), bench_<tag>.json (the default bench.py line).
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh $TAG B > gpurun_out/collect.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh $TAG B 1 > gpurun_out/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
if [ -f vampire_amd/_lib/abl_mstamps.so ]; then
  VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_mstamps.so python tools/debug/merged_stamps.py B 1 gpurun_out/merged_timeline_$TAG.txt > gpurun_out/stamps.log 2>&1
fi
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest_gpu.log | tail -3
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_$TAG.json; cut -c1-300 gpurun_out/bench_$TAG.json
