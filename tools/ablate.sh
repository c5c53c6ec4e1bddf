#!/bin/bash
# Build variants of one source file with -D flags into separate shared libraries (here, no GPU).
# usage: tools/ablate.sh <file.hip> <name1>=<flags> ...   -> vampire_amd/_lib/abl_<name>.so
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$1; shift
BASE=$(basename $SRC .hip)
OBJS=$(ls $ROOT/vampire_amd/_lib/*.o | grep -v "/$BASE.o" | grep -v abl_)
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 $flags -c $ROOT/vampire_amd/csrc/$BASE.hip -o $ROOT/vampire_amd/_lib/abl_${name}_$BASE.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/vampire_amd/_lib/abl_$name.so $OBJS $ROOT/vampire_amd/_lib/abl_${name}_$BASE.o
  echo built abl_$name.so
done
