#!/bin/bash
# GPU box: the replayed training step under HIP-runtime settings that touch graph execution / cross-queue signalling
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "== $*: $(env "$@" timeout 90 python3 $ROOT/tools/try_graph.py B 1 200 2>&1 | grep 'graph\|rror' | head -2 | tr '\n' ' ')"; }
run A=0
# (ROC_SYSTEM_SCOPE_SIGNAL=0 hangs the replay on this image: do not run it)
run GPU_STREAMOPS_CP_WAIT=0
run GPU_STREAMOPS_CP_WAIT=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run A=0
