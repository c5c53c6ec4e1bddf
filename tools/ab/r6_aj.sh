#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in "" nb4 nb3 nb2; do
  if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
  echo "== ${v:-nb8}: B1 $(python tools/try_graph.py B 1 300 2>&1 | grep -i graph) | B8 $(python tools/try_graph.py B 8 100 2>&1 | grep -i graph) | $(python tools/debug/bs_kernels.py quick 2>&1 | grep -E 'lift_bwd_fill')"
done
done
