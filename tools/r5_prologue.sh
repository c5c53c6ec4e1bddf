#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in default "$@"; do
  rm -rf $ROOT/gpurun_out/kt
  if [ $v != default ]; then export VAMPIRE_HIP_LIB=$ROOT/vampire_amd/_lib/abl_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/kt -- python3 $ROOT/tools/r5_prologue.py > /dev/null 2>&1
  echo "== $v"
  python3 - <<PY
import csv, glob
for f in glob.glob("$ROOT/gpurun_out/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "vamp" in r["Name"]: print("%-80s %5s %9.1f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
