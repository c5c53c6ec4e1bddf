#!/bin/bash
# Run ON THE GPU BOX: kernel table of a script under each ablation library built by tools/ablate.sh.
# usage: tools/abl_run.sh <rows> "<name1> <name2> ..." <script.py> [args...]   ("base" = the product library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROWS=$1; shift
NAMES=$1; shift
for n in $NAMES; do
  echo "== $n"
  if [ "$n" = base ]; then unset VAMPIRE_HIP_LIB; else export VAMPIRE_HIP_LIB=$ROOT/vampire_amd/_lib/abl_$n.so; fi
  bash $ROOT/tools/kstats_cmd.sh $ROWS "$@"
done
