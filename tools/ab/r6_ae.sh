#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "camera or render or step or graph" 2>&1 | grep -E "passed|failed" | tail -2
for r in 1 2; do
for v in "" nopf; do
  if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
  echo "== ${v:-prefetch}: B1 $(python tools/try_graph.py B 1 300 2>&1 | grep -i graph) | B8 $(python tools/try_graph.py B 8 100 2>&1 | grep -i graph) | noert $(python tools/try_graph.py B 1 200 ert=0 2>&1 | grep -i graph)"
done
done
unset VAMPIRE_HIP_LIB
python tools/debug/bs_kernels.py quick 2>&1 | grep -E "kernel|cam_bwd|sum"
