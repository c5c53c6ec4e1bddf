#!/bin/bash
# GPU box: forward pair and replayed step with the density samples on the reference chain (1) / the ray line (0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2 3; do
  for x in 0 1; do
    echo "== exact=$x: $(VAMP_CAM_EXACT=$x python3 $ROOT/tools/fwd_graph.py B 1 200 0 2>&1 | grep 'forward pair') | $(VAMP_CAM_EXACT=$x python3 $ROOT/tools/try_graph.py B 1 200 2>&1 | grep graph)"
  done
done
