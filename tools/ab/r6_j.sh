#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for B in 2 4 8; do
  echo "B=$B overlap=1: $(VAMP_OVERLAP=1 python tools/try_graph.py B $B 60 2>&1 | grep -i 'graph')"
  echo "B=$B overlap=0: $(VAMP_OVERLAP=0 python tools/try_graph.py B $B 60 2>&1 | grep -i 'graph')"
done; done
