#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -k "merged or render_forward_backward or debug_checks or graph_capturable or variants or backbone or splat" 2>&1 | grep -v GridwiseOp | tail -3
for r in 1 2 3; do
  echo "swap=1: $(python tools/try_graph.py B 1 200 bwd_swap=1 2>&1 | grep -i 'graph')"
  echo "swap=0: $(python tools/try_graph.py B 1 200 bwd_swap=0 2>&1 | grep -i 'graph')"
done
