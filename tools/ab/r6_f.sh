#!/bin/bash
cd $GRAFT_REPO_ROOT
VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_mix3.so timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "merged" 2>&1 | tail -3
for r in 1 2 3; do
  for v in "" mix2 mix3 mix5 mix8; do
    if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
    echo "== ${v:-default(mix0)}: $(python tools/fwd_graph.py B 1 300 0 2>&1 | grep 'forward pair')"
  done
done
