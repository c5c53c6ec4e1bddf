#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python3 tools/try_graph.py B 1 50 ert=0 > /dev/null 2>&1
python3 tools/debug/graph_timeline.py $(find /tmp/tl -name "p_kernel_trace.csv")
