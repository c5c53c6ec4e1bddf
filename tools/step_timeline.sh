#!/bin/bash
# GPU box: kernel timeline of one replayed training-step graph
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/stl
python3 $ROOT/tools/try_graph.py ${1:-B} 1 200 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/stl -o p -- python3 $ROOT/tools/try_graph.py ${1:-B} 1 30 > /dev/null 2>&1
python3 $ROOT/tools/debug/graph_timeline.py $(ls $ROOT/gpurun_out/stl/*/p_kernel_trace.csv $ROOT/gpurun_out/stl/p_kernel_trace.csv 2>/dev/null | head -1)
