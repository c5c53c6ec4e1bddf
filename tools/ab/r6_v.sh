#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python bench.py 2>&1 | tail -1 > gpurun_out/bench_new.json; cat gpurun_out/bench_new.json
