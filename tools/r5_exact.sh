#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2 3; do
  for x in 0 1; do
    echo "== exact=$x: $(VAMP_CAM_EXACT=$x python3 $ROOT/tools/fwd_graph.py B 1 200 0 2>&1 | grep 'forward pair') | $(VAMP_CAM_EXACT=$x python3 $ROOT/tools/try_graph.py B 1 200 2>&1 | grep graph)"
  done
done
