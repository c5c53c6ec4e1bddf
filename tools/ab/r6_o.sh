#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for B in 2 3 4; do
  echo "B=$B merged=1: $(python tools/try_graph.py B $B 100 fwd_merged=1 2>&1 | grep -i 'graph')"
  echo "B=$B merged=0: $(python tools/try_graph.py B $B 100 fwd_merged=0 2>&1 | grep -i 'graph')"
done; done
