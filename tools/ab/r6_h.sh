#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -k "lift or full_size_gradients or elementwise or cfg_d or backbone or graph_capturable or edge_cases" 2>&1 | grep -v GridwiseOp | tail -6
for r in 1 2; do python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"; done
bash tools/kstats.sh B 1 8
