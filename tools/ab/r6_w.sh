#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/debug/fwd_flags.py 2>&1 | tail -4
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
