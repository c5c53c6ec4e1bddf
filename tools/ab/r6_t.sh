#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -k "camera or render or variants or splat or step or graph" 2>&1 | tail -3
for v in "" ch128 ch512 chinf; do
  if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
  echo "== ${v:-default}"
  python tools/debug/bs_kernels.py quick 2>&1 | grep -E "cam_bwd_heavy|cam_bwd_gather|aux"
  for r in 1 2; do python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"; done
  python tools/try_graph.py B 8 100 2>&1 | grep -i "graph"
  python tools/try_graph.py B 1 200 ert=0 2>&1 | grep -i "graph"
done
