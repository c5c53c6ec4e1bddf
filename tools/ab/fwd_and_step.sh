#!/bin/bash
# GPU box: A/B of library variants on the no-grad forward pair (one-stream graph) and the replayed training step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
    echo "== ${v:-default}: $(python3 $ROOT/tools/fwd_graph.py B 1 200 0 2>&1 | grep 'forward pair') | $(python3 $ROOT/tools/try_graph.py B 1 200 2>&1 | grep graph)"
  done
done
