#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  export VAMP_X_CAM_FIRST=$v
  echo "== cam first $v"
  for r in 1 2; do python tools/try_graph.py B 1 300 2>&1 | grep -i "graph"; done
  python tools/try_graph.py B 8 100 2>&1 | grep -i "graph"
done
export VAMP_X_CAM_FIRST=1
export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python3 tools/try_graph.py B 1 50 > /dev/null 2>&1
python3 tools/debug/graph_timeline.py $(find /tmp/tl -name "p_kernel_trace.csv")
