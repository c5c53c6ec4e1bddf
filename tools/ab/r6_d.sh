#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "lift or merged or camera_direct or elementwise or direct_kernel_taps or smooth" 2>&1 | tail -4
echo "--- feat_cl=1"; bash tools/kstats_cmd.sh 3 tools/fwd_graph.py B 1 100 0 feat_cl=1
for r in 1 2; do
python tools/fwd_graph.py B 1 200 0 feat_cl=0 2>&1 | grep 'forward pair'
python tools/fwd_graph.py B 1 200 0 feat_cl=1 2>&1 | grep 'forward pair'
done
