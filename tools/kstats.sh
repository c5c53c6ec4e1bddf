#!/bin/bash
# Run ON THE GPU BOX: per-kernel average durations of one training step (rocprofv3 kernel trace).
# usage: tools/kstats.sh [cfg] [batch] [rows]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/kt -- python3 $ROOT/tools/time_bwd.py ${1:-B} ${2:-1} > /dev/null 2>&1
cd $ROOT
python3 - ${3:-24} <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/kt/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:int(sys.argv[1])]:
        print("%-72s %5s %9.1f" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
