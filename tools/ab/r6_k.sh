#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -k "lift or full_size_gradients or elementwise or graph_capturable or backbone or debug_checks" 2>&1 | grep -v GridwiseOp | tail -3
for r in 1 2; do python tools/fwd_graph.py B 1 300 0 2>&1 | grep 'forward pair'; done
for r in 1 2; do python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"; done
python tools/try_graph.py B 8 60 2>&1 | grep -i "graph"
bash tools/kstats_cmd.sh 3 tools/fwd_graph.py B 1 100 0
