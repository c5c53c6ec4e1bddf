#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "step or graph or merged or render_forward" 2>&1 | grep -E "passed|failed" | tail -2
for r in 1 2 3; do
  echo "B1: $(python tools/try_graph.py B 1 300 2>&1 | grep -i graph)"
done
export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python3 tools/try_graph.py B 1 50 > /dev/null 2>&1
python3 tools/debug/graph_timeline.py $(find /tmp/tl -name "p_kernel_trace.csv")
