#!/bin/bash
# GPU box: A/B of library variants on the no-grad forward pair with early termination OFF (copy + planned march)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
    echo "== ${v:-default} ERT off: $(VAMP_ERT=0 python3 $ROOT/tools/fwd_graph.py B 1 100 0 2>&1 | grep 'forward pair')"
  done
done
