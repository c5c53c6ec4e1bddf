#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  export VAMP_X_FORK=1; echo "fork after ray issue: $(python tools/try_graph.py B 1 300 2>&1 | grep -i graph)"
  export VAMP_X_FORK=0; echo "fork first:           $(python tools/try_graph.py B 1 300 2>&1 | grep -i graph)"
done
export VAMP_X_FORK=1
export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python3 tools/try_graph.py B 1 50 > /dev/null 2>&1
python3 tools/debug/graph_timeline.py $(find /tmp/tl -name "p_kernel_trace.csv") | sed -n 4,8p
