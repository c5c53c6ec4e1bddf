#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -k "ert or termination or regime or step or graph or variants" 2>&1 | grep -E "passed|failed" | tail -2
for r in 1 2; do
  echo "B1 $(python tools/try_graph.py B 1 300 2>&1 | grep -i graph) | noert $(python tools/try_graph.py B 1 200 ert=0 2>&1 | grep -i graph) | naive $(python tools/try_graph.py B 1 200 cam_direct=0 2>&1 | grep -i graph)"
done
