#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "merged or checks_the_heights or camera_direct_forward or elementwise or direct_kernel_taps or render_forward_backward_tiny or graph_capturable" 2>&1 | tail -15
for r in 1 2; do
  python tools/fwd_graph.py B 1 200 0 fwd_merged=0 2>&1 | grep "forward pair"
  python tools/fwd_graph.py B 1 200 0 fwd_merged=1 2>&1 | grep "forward pair"
done
for r in 1 2; do
  VAMP_X_TRAIN_MERGED=0 python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"
  VAMP_X_TRAIN_MERGED=1 python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"
done
VAMPIRE_HIP_LIB=vampire_amd/_lib/abl_mstamps.so python tools/debug/merged_stamps.py B 1 gpurun_out/merged_stamps_B1.txt
