#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  export VAMP_X_BEV_FIRST=0; echo "bev after march: $(python tools/try_graph.py B 1 200 ert=0 2>&1 | grep -i graph)"
  export VAMP_X_BEV_FIRST=1; echo "bev beside copy: $(python tools/try_graph.py B 1 200 ert=0 2>&1 | grep -i graph)"
done
