#!/bin/bash
# GPU box: A/B of library variants on the replayed training step (ERT on) + the kernel's duration in the timeline
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
    echo "== ${v:-default}: $(python3 $ROOT/tools/try_graph.py B 1 200 2>&1 | grep graph)"
  done
done
for v in "" "$@"; do
  if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
  echo "== ${v:-default}"; $ROOT/tools/step_timeline.sh B 2>&1 | grep "${KERNEL:-cam_bwd_ray}\|span"
done
