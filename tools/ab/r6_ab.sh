#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|Error" gpurun_out/pytest_gpu.log | tail -5
for r in 1 2 3; do
  echo "B1: $(python tools/try_graph.py B 1 300 2>&1 | grep -i graph)"
done
echo "B8: $(python tools/try_graph.py B 8 100 2>&1 | grep -i graph)"
echo "noert: $(python tools/try_graph.py B 1 200 ert=0 2>&1 | grep -i graph)"
export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python3 tools/try_graph.py B 1 50 > /dev/null 2>&1
python3 tools/debug/graph_timeline.py $(find /tmp/tl -name "p_kernel_trace.csv")
