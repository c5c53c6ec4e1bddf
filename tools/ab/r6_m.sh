#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "lift" 2>&1 | grep -v GridwiseOp | tail -2
bash tools/kstats_cmd.sh 2 tools/fwd_graph.py B 1 200 0
for r in 1 2; do python tools/fwd_graph.py B 1 300 0 2>&1 | grep 'forward pair'; done
