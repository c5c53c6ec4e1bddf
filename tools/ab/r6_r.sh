#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -k "camera or render or variants or splat or step or graph" 2>&1 | tail -3
for r in 1 2; do python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"; done
python tools/try_graph.py B 8 100 2>&1 | grep -i "graph"
python tools/try_graph.py B 1 200 ert=0 2>&1 | grep -i "graph"
python tools/debug/bs_kernels.py quick 2>&1 | grep -E "kernel|cam_bwd|aux|sum"
echo "--- kernels B 1"; bash tools/kstats_cmd.sh 13 tools/try_graph.py B 1 100
echo "--- kernels B 1 ert=0"; bash tools/kstats_cmd.sh 14 tools/try_graph.py B 1 100 ert=0
