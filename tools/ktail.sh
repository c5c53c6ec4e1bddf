#!/bin/bash
# Run ON THE GPU BOX: per-kernel time of the LAST part of a run (steady state, after MIOpen's find
# phase and warm-up): rocprofv3 kernel trace, kernels that start in the last FRAC of the traced
# interval.  usage: tools/ktail.sh <frac> <rows> <script.py> [args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
FRAC=$1; ROWS=$2; shift; shift
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/kt
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/kt -- python3 $SCRIPT "$@" > /dev/null 2>&1
cd $ROOT
python3 - $FRAC $ROWS <<'PY'
import csv, glob, sys
frac, rows = float(sys.argv[1]), int(sys.argv[2])
for f in glob.glob("gpurun_out/kt/*/*kernel_trace.csv"):
    rs = list(csv.DictReader(open(f)))
    t0 = min(int(r["Start_Timestamp"]) for r in rs); t1 = max(int(r["End_Timestamp"]) for r in rs)
    cut = t1 - (t1 - t0) * frac
    agg = {}
    for r in rs:
        if int(r["Start_Timestamp"]) < cut: continue
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(r["Kernel_Name"], [0, 0]); a[0] += 1; a[1] += d
    tot = sum(v[1] for v in agg.values())
    print("window %.1f ms, kernel time %.1f ms" % ((t1 - cut) / 1e6, tot / 1e6))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:rows]:
        print("%-100s %6d %9.1f us avg %7.2f ms total" % (k[:100], v[0], v[1] / v[0] / 1e3, v[1] / 1e6))
PY
rm -rf gpurun_out/kt
