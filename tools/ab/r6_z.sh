#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for b in 1 4 8; do
  echo "== B=$b default: $(python tools/try_graph.py B $b 100 2>&1 | grep -i graph)"
  echo "== B=$b merged always: $(python tools/try_graph.py B $b 100 merged_tiles=1000000 2>&1 | grep -i graph)"
done
done
