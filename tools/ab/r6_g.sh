#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -k "render or camera or elementwise or direct_kernel_taps or smooth or cfg_d or regimes or ert or merged or chosen" 2>&1 | tail -6
python tools/debug/cfgd_abs_err.py
for r in 1 2; do
python tools/fwd_graph.py B 1 300 0 2>&1 | grep 'forward pair'
python tools/fwd_graph.py B 1 300 0 ert=0 2>&1 | grep 'forward pair'
done
python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"
bash tools/kstats_cmd.sh 3 tools/fwd_graph.py B 1 100 0
