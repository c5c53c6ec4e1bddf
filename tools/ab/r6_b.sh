#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "lift or merged or checks_the_heights or elementwise or graph_capturable or bev" 2>&1 | tail -8
for r in 1 2; do
  for v in "" mp3 mp5; do
    if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
    echo "== ${v:-default}: $(python tools/fwd_graph.py B 1 200 0 2>&1 | grep 'forward pair')"
  done
done
unset VAMPIRE_HIP_LIB
python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"
python tools/try_graph.py B 8 100 2>&1 | grep -i "graph"
