#!/bin/bash
# GPU box: A/B of library variants (vampire_amd/_lib/abl_<name>.so; "" = the shipped one) on the replayed
# training step with early ray termination on and off.  usage: tools/ab/step_ert.sh name...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
    for e in 1 0; do
      echo "== ${v:-default} ERT=$e: $(VAMP_ERT=$e python3 $ROOT/tools/try_graph.py B 1 200 2>&1 | grep graph)"
    done
  done
done
