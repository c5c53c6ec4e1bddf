#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in "" noshare share_p2 noshare_p2; do
    if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
    echo "== ${v:-default(share,p3)}: $(python tools/fwd_graph.py B 1 300 0 2>&1 | grep 'forward pair')"
  done
done
