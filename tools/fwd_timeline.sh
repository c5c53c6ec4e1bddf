#!/bin/bash
# GPU box: kernel timeline of a replayed forward-pair graph (two streams / one stream)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ov in 1 0; do
  rm -rf $ROOT/gpurun_out/ftl$ov
  python3 $ROOT/tools/fwd_graph.py ${1:-B} 1 100 $ov 2>&1 | grep -v amdgpu.ids
  rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/ftl$ov -o p -- python3 $ROOT/tools/fwd_graph.py ${1:-B} 1 30 $ov > /dev/null 2>&1
  python3 $ROOT/tools/debug/graph_timeline.py $(ls $ROOT/gpurun_out/ftl$ov/*/p_kernel_trace.csv $ROOT/gpurun_out/ftl$ov/p_kernel_trace.csv 2>/dev/null | head -1)
done
