#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in t16 "" t16 "" ; do
  if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
  echo "== ${v:-default}"
  for r in 1 2; do python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"; done
  python tools/try_graph.py B 8 100 2>&1 | grep -i "graph"
  python tools/try_graph.py B 1 200 ert=0 2>&1 | grep -i "graph"
done
