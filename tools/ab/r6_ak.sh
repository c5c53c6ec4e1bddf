#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" nostore noatomic neither; do
  if [ -n "$v" ]; then export VAMPIRE_HIP_LIB=$GRAFT_REPO_ROOT/vampire_amd/_lib/abl_$v.so; else unset VAMPIRE_HIP_LIB; fi
  echo "== ${v:-default}: $(python tools/debug/bs_kernels.py quick 2>&1 | grep -E 'lift_bwd_fill')"
done
