#!/bin/bash
# GPU box: lift forward after the cull words (round 5): parity subset, per-kernel times, SQ counters
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "lift or full_size_checksums or cfg_d or edge_cases or full_size_properties or full_size_elementwise" 2>&1 | grep -v GridwiseOp | tail -8
echo "== time_lift B 1"; python tools/time_lift.py B 1
echo "== time_lift B 1 logits"; python tools/time_lift.py B 1 logits
echo "== time_kernels B"; python tools/time_kernels.py --cfg B
echo "== time_kernels D"; python tools/time_kernels.py --cfg D
