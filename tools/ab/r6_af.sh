#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|Error" gpurun_out/pytest_gpu.log | tail -3
for r in 1 2 3; do
  echo "B1 $(python tools/try_graph.py B 1 300 2>&1 | grep -i graph) | B8 $(python tools/try_graph.py B 8 100 2>&1 | grep -i graph) | noert $(python tools/try_graph.py B 1 200 ert=0 2>&1 | grep -i graph)"
done
