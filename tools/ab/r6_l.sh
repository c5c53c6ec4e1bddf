#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
python tools/fwd_graph.py B 1 200 0 ert=0 cam_direct=0 2>&1 | grep 'forward pair'
python tools/fwd_graph.py B 1 200 0 ert=0 cam_direct=1 2>&1 | grep 'forward pair'
echo "step ert=0 planned: $(VAMP_ERT=0 VAMP_CAM_DIRECT=0 python tools/try_graph.py B 1 100 2>&1 | grep -i graph)"
echo "step ert=0 direct : $(VAMP_ERT=0 VAMP_CAM_DIRECT=1 python tools/try_graph.py B 1 100 2>&1 | grep -i graph)"
done
