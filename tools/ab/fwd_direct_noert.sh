#!/bin/bash
# GPU box: the one-kernel camera forward forced (VAMP_CAM_DIRECT=1) without early termination: forward pair of variants
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
    echo "== ${v:-default} ERT off, direct: $(VAMP_ERT=0 VAMP_CAM_DIRECT=1 python3 $ROOT/tools/fwd_graph.py B 1 100 0 2>&1 | grep 'forward pair')"
  done
done
