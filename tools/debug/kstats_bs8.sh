#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/ks8
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ks8 -- python3 $ROOT/tools/try_graph.py B 8 30 > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$ROOT/gpurun_out/ks8/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:22]:
        print("%-70s calls %5s avg %8.1f us  per-sample %7.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["AverageNs"]) / 8e3))
PY
rm -rf $ROOT/gpurun_out/ks8
