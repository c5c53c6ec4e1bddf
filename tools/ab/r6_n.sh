#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -k "merged or render or camera or debug_checks or graph_capturable or elementwise or regimes or ert or variants or backbone or chosen" 2>&1 | grep -v GridwiseOp | tail -5
for r in 1 2; do python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"; done
python tools/try_graph.py B 8 60 2>&1 | grep -i "graph"
