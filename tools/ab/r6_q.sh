#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -k "camera or render or variants or splat or step or graph" 2>&1 | tail -8
python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"
python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"
python tools/try_graph.py B 8 100 2>&1 | grep -i "graph"
python tools/try_graph.py B 1 200 ert=0 2>&1 | grep -i "graph"
echo "--- kernels B 1"; bash tools/kstats_cmd.sh 12 tools/try_graph.py B 1 100
