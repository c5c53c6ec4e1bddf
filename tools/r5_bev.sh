#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "render or bev or checksums or edge" 2>&1 | grep -v GridwiseOp | tail -3
for v in default "$@"; do
  if [ $v != default ]; then export VAMPIRE_HIP_LIB=$ROOT/vampire_amd/_lib/abl_$v.so; fi
  echo "== $v"; bash tools/r5_fwdtl.sh B 2>&1 | grep -v "^ *[0-9.]* *[0-9.]* *[0-9]  void vamp::lift\|start_us" 
  python tools/time_bwd.py B 1 2>&1 | grep -v amdgpu
done
