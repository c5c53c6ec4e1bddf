#!/bin/bash
# Run ON THE GPU BOX: A/B of library variants (tools/ablate.sh) with the default bench, 2 rounds.
# usage: tools/ab_lib.sh "" base p1n8 ...      ("" = the default library)
for round in 1 2; do
for v in "$@"; do
  echo -n "[$v]  "
  if [ -z "$v" ]; then python bench.py --no-extra --no-cpu-baseline --no-matrix --steps 300 --warmup 20 2>/dev/null | python tools/bq.py | head -1
  else VAMPIRE_HIP_LIB=vampire_amd/_lib/abl_$v.so python bench.py --no-extra --no-cpu-baseline --no-matrix --steps 300 --warmup 20 2>/dev/null | python tools/bq.py | head -1; fi
done; done
