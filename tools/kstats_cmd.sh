#!/bin/bash
# Run ON THE GPU BOX: per-kernel average durations (rocprofv3 kernel trace) of an arbitrary python
# script.  usage: tools/kstats_cmd.sh <rows> <script.py> [args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROWS=$1; shift
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/kt -- python3 $SCRIPT "$@" > /dev/null 2>&1
cd $ROOT
python3 - $ROWS <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/kt/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:int(sys.argv[1])]:
        print("%-90s %5s %9.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
