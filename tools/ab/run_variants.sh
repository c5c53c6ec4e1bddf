#!/bin/bash
# GPU box: time the ablation libraries built by tools/ablate.sh (usage: tools/ab/run_variants.sh <script> <args> -- name...)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
CMD=()
while [ "$1" != "--" ]; do CMD+=("$1"); shift; done
shift
echo "== default"; python "${CMD[@]}" 2>&1 | grep -v amdgpu.ids
for n in "$@"; do
  echo "== $n"; VAMPIRE_HIP_LIB=$ROOT/vampire_amd/_lib/abl_$n.so python "${CMD[@]}" 2>&1 | grep -v amdgpu.ids
done
