#!/bin/bash
# Run ON THE GPU BOX: A/B of environment switches with the default bench, 200 steps each, 2 rounds.
# usage: tools/ab.sh "VAR=a" "VAR=b" ...
for round in 1 2; do
for v in "$@"; do
  echo -n "$v  "
  env $v python bench.py --no-cpu-baseline --steps 200 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f samples/s  %.4f ms' % (d['value'], d['ms_per_step']))"
done; done
